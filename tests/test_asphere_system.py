"""The reference's "Complex aspherical imaging system" KAT (test/runtests.jl:1581-1696): a three-element even-asphere phone-camera
objective + filter + cover glass built with Lens(front_surface, back_surface, thickness, n); three parallel rays must cross the
axis 0.12 mm behind the last surface to 1e-7 m, and the levelling rings have the stated thicknesses.  Exercises the aspheric SDFs
(numeric normals, inexact-distance flag), the ring/levelling logic of the lens builder and the extended-shape kernel variants."""
import numpy as np
import pytest

import bmo_amd as bmo


def build():
    k = 1e3
    E = bmo.EvenAsphericalSurface
    L1 = bmo.Lens(E(1.054e-3, 1.333024e-3, -0.14294, [0, 0.038162 * k ** 3, 0.06317 * k ** 5, -0.020792 * k ** 7, 0.18432 * k ** 9, -0.04827 * k ** 11,
                                                      0.094529 * k ** 13]),
                  E(2.027e-3, 1.216472e-3, 8.0226, [0, 0.0074974 * k ** 3, 0.064686 * k ** 5, 0.19354 * k ** 7, -0.50703 * k ** 9, -0.34529 * k ** 11,
                                                    5.9938 * k ** 13]),
                  0.72e-3, lambda lam: 1.580200)
    L2 = bmo.Lens(E(-3.116e-3, 1.4e-3, -49.984, [0, -0.31608 * k ** 3, 0.34755 * k ** 5, -0.17102 * k ** 7, -0.41506 * k ** 9, -1.342 * k ** 11,
                                                 5.0594 * k ** 13, -2.7483 * k ** 15]),
                  E(-4.835e-3, 1.9e-3, 1.6674, [0, -0.079727 * k ** 3, 0.13899 * k ** 5, -0.044057 * k ** 7, -0.019369 * k ** 9, 0.016993 * k ** 11,
                                                0.093716 * k ** 13, -0.080329 * k ** 15]),
                  0.55e-3, lambda lam: 1.804700)
    bmo.translate3d(L2, [0, L1.thickness + 0.39e-3, 0])
    L3 = bmo.Lens(E(3.618e-3, 3.04e-3, -44.874, [0, -0.14756 * k ** 3, 0.035194 * k ** 5, -0.0032262 * k ** 7, 0.0018592 * k ** 9, 0.00036658 * k ** 11,
                                                 -0.00016039 * k ** 13, -3.1846e-5 * k ** 15]),
                  E(2.161e-3, 3.7e-3, -10.719, [0, -0.096568 * k ** 3, 0.026771 * k ** 5, -0.011261 * k ** 7, 0.0019879 * k ** 9, 0.00015579 * k ** 11,
                                                -0.00012433 * k ** 13, 1.5264e-5 * k ** 15]),
                  0.7e-3, lambda lam: 1.580200)
    bmo.translate_to3d(L3, L2.position())
    bmo.translate3d(L3, [0, L2.thickness + 0.63e-3, 0])
    Filt = bmo.Lens(bmo.CircularFlatSurface(4.2e-3), 0.15e-3, lambda lam: 1.516800)
    bmo.translate_to3d(Filt, L3.position())
    bmo.translate3d(Filt, [0, L3.thickness + 0.19e-3, 0])
    Cover = bmo.Lens(bmo.CircularFlatSurface(4.9e-3), 0.5e-3, lambda lam: 1.469200)
    bmo.translate_to3d(Cover, Filt.position())
    bmo.translate3d(Cover, [0, Filt.thickness + 0.18e-3, 0])
    return L1, L2, L3, Filt, Cover


def run_kat(solve):
    L1, L2, L3, Filt, Cover = build()
    for lens, t in ((L1, 0.72e-3), (L2, 0.55e-3), (L3, 0.7e-3), (Filt, 0.15e-3), (Cover, 0.5e-3)):
        assert np.isclose(lens.thickness, t, rtol=1.5e-8)
    system = bmo.System([L1, L2, L3, Filt, Cover])
    for z in (-1.3e-3 / 2, 0.0, 1.3e-3 / 2):
        beam = bmo.Beam([0, -0.5e-3, z], [0, 1, 0], 0.5876e-6)
        solve(system, beam, 50)
        last = beam.rays[-1]
        f_pos = last.pos + 0.12e-3 * last.dir
        assert abs(f_pos[2]) <= 1e-7
    assert abs(2 * L1.shape.sdfs[3].hthickness - 0.00060839) <= 1e-6
    assert abs(2 * L2.shape.sdfs[3].hthickness - 0.00057497) <= 1e-6
    assert abs(2 * L3.shape.sdfs[3].hthickness - 0.00048395) <= 1e-6


def test_kat_aspheric_imaging_system_oracle(oracle):
    run_kat(lambda system, beam, r_max: oracle.solve_system(system, beam, r_max=r_max))


@pytest.mark.gpu
def test_kat_aspheric_imaging_system_engine():
    run_kat(lambda system, beam, r_max: bmo.solve_system(system, beam, r_max=r_max))
