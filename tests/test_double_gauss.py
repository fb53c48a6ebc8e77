"""The reference's "Double Gauss lens" KAT (test/runtests.jl:2693-2762): a six-element objective built from SphericalLens /
SphericalDoubletLens, moved as nested ObjectGroups, imaged onto a Spotdetector by a collimated bundle and two point sources.
Pins the host scene builders (doublets, nested group kinematics, reset functions, the bundle sources) together
with the trace: every hit must land within the stated distance of the detector origin."""
import math

import numpy as np
import pytest

import bmo_amd as bmo


def build():
    l1 = bmo.SphericalLens(48.88e-3, 182.96e-3, 8.89e-3, 52.3e-3, lambda lam: 1.62286)
    l23 = bmo.SphericalDoubletLens(36.92e-3, math.inf, 23.06e-3, 15.11e-3, 2.31e-3, 45.11e-3, lambda lam: 1.58565, lambda lam: 1.67764)
    l45 = bmo.SphericalDoubletLens(-23.91e-3, math.inf, -36.92e-3, 1.92e-3, 7.77e-3, 40.01e-3, lambda lam: 1.57046, lambda lam: 1.64128)
    l6 = bmo.SphericalLens(1063.24e-3, -48.88e-3, 6.73e-3, 45.11e-3, lambda lam: 1.62286)
    l_23 = l1.thickness + 0.38e-3
    l_45 = l_23 + l23.thickness + 9.14e-3 + 13.36e-3
    l_6 = l_45 + l45.thickness + 0.38e-3
    df = 7e-4
    f_z = l_6 + l6.thickness + 58.21e-3 + df
    bmo.translate3d(l23, [0, l_23, 0])
    bmo.translate3d(l45, [0, l_45, 0])
    bmo.translate3d(l6, [0, l_6, 0])
    double_gauss = bmo.ObjectGroup([l1, l23, l45, l6])
    detector = bmo.Spotdetector(5e-3)
    bmo.translate3d(detector, [0, f_z, 0])
    test_setup = bmo.ObjectGroup([double_gauss, detector])
    bmo.translate3d(test_setup, [0.05, 0.05, 0.05])
    bmo.xrotate3d(test_setup, math.radians(60))
    bmo.zrotate3d(test_setup, math.radians(45))
    return bmo.System([test_setup]), test_setup, double_gauss, detector, l1


def coma_ok(detector, atol):
    return bool((np.linalg.norm(detector.data, axis=1) <= atol).all())


def run_kat(solve):
    system, test_setup, double_gauss, detector, l1 = build()
    # collimated bundle along the rotated optical axis: back focal length of the published design (:2734-2740)
    d = double_gauss.orientation()[:, 1]
    pos = l1.position() - 0.05 * d
    src = bmo.CollimatedSource(pos, d, 0.04, 486.0e-9, num_rays=1000, num_rings=10)
    solve(system, src)
    assert len(detector.data) == len(src.beams)
    assert coma_ok(detector, 2e-5)
    # back to the origin, y-axis alignment (:2742-2745)
    bmo.reset_rotation3d(test_setup)
    bmo.reset_translation3d(test_setup)
    bmo.translate_to3d(detector, [0, 0.147, 0])
    for theta, atol in ((math.radians(2), 6e-5), (5e-5, 2e-7)):  # wide (:2748-2753) and narrow (:2755-2761, issue 11) point source
        src = bmo.PointSource([0, -0.5, 0], [0, 1, 0], theta, 486.0e-9, num_rays=1000, num_rings=10)
        detector.empty()
        solve(system, src)
        assert coma_ok(detector, atol)
        # All 1000 rays arrive, the chief ray included: it runs exactly along the axis and leaves the second doublet through the
        # apex of a concave surface, where the cylinder part of ConcaveSphericalSurfaceSDF takes norm() of a zero vector.  This KAT
        # is what pins the max / min rule of the dual numbers (oracle/jl_math.hpp): if the NaN partials of that norm leak through the
        # sdf's outer max (DiffRules' product form), every concave normal falls back to central differences, the narrow bundle (all
        # rays within 25 um of the apexes) lands 1e-3 m off instead of 2e-7 m; and if ties go to the first argument the chief ray gets a
        # NaN normal (refraction3d would throw in the reference).
        lost = [i for i, b in enumerate(src.beams) if b.rays[-1].intersection is None or b.rays[-1].intersection.object is not detector]
        assert lost == [], lost
        assert len(detector.data) == len(src.beams)


def test_kat_double_gauss_oracle(oracle):
    run_kat(lambda system, beams: oracle.solve_system(system, beams, threads=8))


@pytest.mark.gpu
def test_kat_double_gauss_engine():
    run_kat(bmo.solve_system)
