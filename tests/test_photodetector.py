"""Photodetector field accumulation: interact3d(::Photodetector, ::GaussianBeamlet, ray_id)
src/OpticalComponents/Detectors/Photodetector.jl:69-107, electric_field(gauss, r, z) src/Gaussian.jl:381-392.

CPU part: the oracle reproduces the reference's interference KATs (test/runtests.jl:1974-2068 two-beam fringes and lambda
sweep, :2070-2165 Michelson interferometer with retracing, :2168-2215 power conservation on tilted detectors).
GPU part: bmo_photodetector_field against the oracle on the same solved batch.
"""
import math

import numpy as np
import pytest

import bmo_amd as bmo
import scenes
from parity import compare, emu_trace

Z0 = bmo.linalg.Z_vacuum
mm = 1e-3


# analytic TEM00 field, OpticUtils.jl:55-97
def efield_analytic(r, z, E0, w0, lam, M2=1):
    zr = math.pi * w0 ** 2 / lam / M2
    w = w0 * np.sqrt(1 + (z / zr) ** 2)
    k = 2 * math.pi / lam
    psi = -np.arctan(z / zr)
    R = z / (z ** 2 + zr ** 2)
    return E0 * w0 / w * np.exp(-r ** 2 / w ** 2) * np.exp(1j * (k * z + psi + (k * r ** 2 * R) / 2))


def intensity(E):
    return np.abs(E) ** 2 / (2 * Z0)


# ------------------------------------------------------------------------------------------------ reference KATs (oracle)
def test_kat_two_beam_fringe_pattern(oracle):  # runtests.jl:1974-2035
    w0, lam, M2, P0 = 0.01e-3, 1000e-9, 1, 1e-3
    E0 = math.sqrt(2 * (2 * P0 / (math.pi * w0 ** 2)) * Z0)
    z, l, n = 0.1, 1e-2, 1000
    pd = bmo.Photodetector(l, n)
    bmo.translate3d(pd, [0, z, 0])
    system = bmo.System([pd])
    dz = 5e-3
    xs = bmo.linalg.linrange(-l / 2, l / 2, n)
    r = np.sqrt(xs[:, None] ** 2 + xs[None, :] ** 2)
    screen = efield_analytic(r, z, E0, w0, lam, M2) + efield_analytic(r, z + dz, E0, w0, lam, M2)
    pd.empty()
    g1 = bmo.GaussianBeamlet([0.0, 0, 0], [0.0, 1, 0], lam, w0, M2=M2, P0=P0)
    g2 = bmo.GaussianBeamlet([0.0, -dz, 0], [0.0, 1, 0], lam, w0, M2=M2, P0=P0)
    oracle.solve_system(system, g1)
    oracle.solve_system(system, g2)
    assert np.all(np.abs(intensity(screen) - pd.intensity()) <= 2e-1)
    assert abs(pd.optical_power() - 2 * P0) <= 3e-5


def test_kat_lambda_phase_shift_through_a_lens(oracle):  # runtests.jl:2037-2068
    w0, lam, M2, P0 = 0.01e-3, 1000e-9, 1, 1e-3
    z, l, n = 0.1, 1e-2, 1000
    R1 = R2 = d = 0.01
    nl = 1.5
    f = 1 / ((nl - 1) * (1 / R1 - 1 / -R2))  # lensmakers_eq(R1, -R2, nl)
    pd = bmo.Photodetector(l / 10, n // 10)
    ln = bmo.ThinLens(R1, R2, d, nl)
    bmo.translate3d(pd, [0, z, 0])
    bmo.translate3d(ln, [0, z - f - ln.thickness / 2, 0])
    system = bmo.System([pd, ln])
    dzs = bmo.linalg.linrange(0, lam, 50)
    P_num = np.zeros(len(dzs))
    for i, zi in enumerate(dzs):
        pd.empty()
        g1 = bmo.GaussianBeamlet([0.0, 0, 0], [0.0, 1, 0], lam, w0, M2=M2, P0=P0)
        g2 = bmo.GaussianBeamlet([0.0, zi, 0], [0.0, 1, 0], lam, w0, M2=M2, P0=P0)
        oracle.solve_system(system, g1)
        oracle.solve_system(system, g2)
        P_num[i] = pd.optical_power()
        assert math.isclose(g1.length(), z, rel_tol=1.5e-8)
        assert math.isclose(g1.length(), g1.optical_path_length() - ln.thickness * (nl - 1), rel_tol=1.5e-8)
        assert math.isclose(g2.length(), g1.length() - zi, rel_tol=1.5e-8)
    P_ana = 4 * P0 * (np.cos(2 * math.pi * dzs / dzs.max()) + 1) / 2
    assert np.all(np.abs(P_num - P_ana) <= 1e-4)


def michelson(pd_resolution=100):
    l_0 = 0.1
    pd_size = bmo.inch / 5
    m1 = bmo.SquarePlanoMirror2D(bmo.inch)
    m2 = bmo.SquarePlanoMirror2D(bmo.inch)
    bs = bmo.ThinBeamsplitter(bmo.inch, reflectance=0.5)
    pd = bmo.Photodetector(pd_size, pd_resolution)
    bmo.translate3d(m1, [l_0, 0, 0])
    bmo.translate3d(m2, [0, l_0, 0])
    bmo.translate3d(pd, [-l_0, 0, 0])
    bmo.zrotate3d(bs, math.radians(45))
    bmo.zrotate3d(m1, math.radians(90))
    bmo.zrotate3d(pd, math.radians(90))
    return bmo.System([m1, m2, bs, pd]), m1, m2, bs, pd, l_0, pd_size


def michelson_equal_arm_sweep(solve, n_steps):
    """runtests.jl:2092-2121: mirror 2 sweeps -lambda..lambda, the same beam is re-solved (retraced) every step."""
    system, m1, m2, bs, pd, l_0, pd_size = michelson()
    assert math.isclose(bs.reflectance ** 2 + bs.transmittance ** 2, 1.0)
    lam, P_0 = 635e-9, 5e-3
    beam = bmo.GaussianBeamlet([0, -l_0, 0], [0, 1.0, 0], lam, 1e-4, P0=P_0)
    lambdas = bmo.linalg.linrange(-lam, lam, n_steps)
    path, pwr = np.zeros(n_steps), np.zeros(n_steps)
    for i, dl in enumerate(lambdas):
        bmo.translate_to3d(m2, np.array([0, l_0, 0]) + np.array([0, dl, 0]))
        pd.empty()
        solve(system, beam)
        path[i] = beam.children[0].children[1].length()
        pwr[i] = pd.optical_power()
    path_ana = 2 * lambdas + 4 * l_0
    pwr_ana = P_0 * (0.5 * np.cos(2 * math.pi * (2 * lambdas / lam) + math.pi) + 0.5)
    assert np.all(np.abs(pwr_ana - pwr) <= 5e-6)
    assert np.allclose(path_ana, path, rtol=1.5e-8, atol=0)
    return beam


def test_kat_michelson_equal_arm_power_with_retracing(oracle):
    michelson_equal_arm_sweep(oracle.solve_system, 200)


def michelson_unequal_arm_field(solve):
    """runtests.jl:2123-2164."""
    system, m1, m2, bs, pd, l_0, pd_size = michelson()
    lam, w0, P0, M2 = 635e-9, 1e-4, 1e-3, 1
    E0 = math.sqrt(2 * (2 * P0 / (math.pi * w0 ** 2)) * Z0) * 1 / math.sqrt(2) ** 2
    beam = bmo.GaussianBeamlet([0, -l_0, 0], [0, 1.0, 0], lam, w0, P0=P0, M2=M2)
    dl = 1 * l_0
    bmo.translate_to3d(m2, [0, l_0 + dl, 0])
    pd.empty()
    solve(system, beam)
    short_arm, long_arm = 4 * l_0, 4 * l_0 + 2 * dl
    xs = bmo.linalg.linrange(-pd_size / 2, pd_size / 2, 100)
    r = np.sqrt(xs[:, None] ** 2 + xs[None, :] ** 2)
    screen = efield_analytic(r, short_arm, E0, w0, lam, M2) + efield_analytic(r, long_arm, E0, w0, lam, M2) * np.exp(1j * math.pi)
    assert np.all(np.abs(screen.real - pd.field.real) <= 5e-2)
    assert np.all(np.abs(screen.imag - pd.field.imag) <= 5e-2)
    return pd


def test_kat_michelson_unequal_arm_field(oracle):
    michelson_unequal_arm_field(oracle.solve_system)


def power_conservation(solve, n_phi=25):
    """runtests.jl:2168-2215: splitter + two detectors with orientation errors; the relative phase of the two inputs is swept by
    rewriting the root E0 and re-solving (retracing)."""
    P0, l0, w0, lam = 0.5, 0.1, 0.5e-3, 1064e-9
    bs = bmo.ThinBeamsplitter(10e-3)
    pd_1 = bmo.Photodetector(10e-3, 100)
    pd_2 = bmo.Photodetector(10e-3, 100)
    bmo.zrotate3d(bs, math.radians(45))
    bmo.translate3d(pd_1, [0, l0, 0])
    bmo.zrotate3d(pd_1, math.radians(180))
    bmo.translate3d(pd_2, [l0, 0, 0])
    bmo.zrotate3d(pd_2, math.radians(90))
    bmo.zrotate3d(bs, math.radians(0.017))
    bmo.zrotate3d(pd_1, math.radians(10))
    bmo.xrotate3d(pd_1, math.radians(15))
    system = bmo.System([bs, pd_1, pd_2])
    l1 = bmo.GaussianBeamlet([0, -l0, 0], [0, 1.0, 0], lam, w0, P0=P0)
    l2 = bmo.GaussianBeamlet([-l0, 0, 0], [1.0, 0, 0], lam, w0, P0=P0)
    E0_buffer = l1.E0
    out = []
    for phi in bmo.linalg.linrange(0, 2 * math.pi, n_phi):
        l1.E0 = E0_buffer * np.exp(1j * phi)
        pd_1.empty()
        pd_2.empty()
        solve(system, l1)
        solve(system, l2)
        p1, p2 = pd_1.optical_power(), pd_2.optical_power()
        assert p1 + p2 - 2 * P0 < 1e-4
        assert p1 + p2 > 2 * P0 - 1e-2  # not in the reference: guards against an empty field passing the one-sided test
        out.append((p1, p2))
    return np.array(out)


def test_kat_power_conservation_tilted_detectors(oracle):
    p = power_conservation(oracle.solve_system)
    assert p[:, 0].max() - p[:, 0].min() > 0.5  # the fringe really moves between the two ports


def test_photodetector_ignores_plain_rays(oracle):  # Photodetector.jl:57-60 (warns, returns nothing)
    pd = bmo.Photodetector(10e-3, 8)
    bmo.translate3d(pd, [0, 0.1, 0])
    beam = bmo.Beam([0, 0, 0], [0, 1, 0], 1e-6)
    oracle.solve_system(bmo.System([pd]), beam)
    assert len(beam.rays) == 1 and beam.rays[0].intersection is not None
    assert not pd.field.any()


# ------------------------------------------------------------------------------------------------ lane code == oracle
def pd_scene(n):
    """C2's train with the transmitted-arm detector replaced by a (tilted) Photodetector; C3's beamlet bundle."""
    objs = scenes.miniscope_objects()
    y0 = 0.332 * mm
    bs = bmo.ThinBeamsplitter(10 * mm)
    bmo.xrotate3d(bs, math.radians(45))
    bmo.translate3d(bs, [0, y0, 30 * mm])
    pd = bmo.Photodetector(5 * mm, 48)
    bmo.xrotate3d(pd, math.radians(90 + 7))
    bmo.zrotate3d(pd, math.radians(4))
    bmo.translate3d(pd, [0, y0, 34 * mm])
    det_r = bmo.Spotdetector(5 * mm)
    bmo.translate3d(det_r, [0, y0 - 4 * mm, 30 * mm])
    system = bmo.System(objs + [bs, pd, det_r])
    return system, pd, scenes.c3_bundle(n)


def test_emulator_records_photodetector_hits_like_the_oracle(oracle):
    system, pd, bundle = pd_scene(64)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    a = oracle.trace(scene, bundle, 20)
    e = emu_trace(scene, bundle, 20)
    compare(e, a, rtol=0.0, label="pd scene")
    assert a.det_count[0] == 3 * 64 and (a.detector_hits(0)[0::3, 0] > 0.9).all()


# ------------------------------------------------------------------------------------------------ HIP engine
def _field_pair(oracle, n, r_max=20):
    system, pd, bundle = pd_scene(n)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    a, osol = oracle.trace(scene, bundle, r_max, threads=16, keep=True)
    g, gsol = bmo.system._engine_solve(scene, bundle, r_max, None)
    compare(g, a, 0.0, "pd scene")
    fa = np.zeros((len(pd.x), len(pd.y)), dtype=np.complex128)
    fg = fa.copy()
    osol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fa)
    ms = gsol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fg)
    return fa, fg, ms, gsol, pd


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 700])
def test_gpu_photodetector_field_matches_oracle(oracle, n):
    fa, fg, ms, gsol, pd = _field_pair(oracle, n)
    peak = np.abs(fa).max()
    assert peak > 0
    assert np.abs(fg - fa).max() <= 1e-9 * peak
    # the call ADDS to the caller's field (the reference accumulates until empty!(pd))
    fg2 = fg.copy()
    gsol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fg2)
    assert np.abs(fg2 - 2 * fg).max() <= 1e-12 * peak


@pytest.mark.gpu
def test_gpu_kat_michelson_sweep_with_retracing():
    beam = michelson_equal_arm_sweep(bmo.solve_system, 24)
    bmo.release(beam)


@pytest.mark.gpu
def test_gpu_kat_michelson_unequal_arm_field(oracle):
    pd_gpu = michelson_unequal_arm_field(bmo.solve_system)
    pd_cpu = michelson_unequal_arm_field(oracle.solve_system)
    assert np.abs(pd_gpu.field - pd_cpu.field).max() <= 1e-9 * np.abs(pd_cpu.field).max()


@pytest.mark.gpu
def test_gpu_kat_power_conservation():
    power_conservation(bmo.solve_system, n_phi=7)


# ------------------------------------------------------------------------------------------------ "Bug fixes" KATs (runtests.jl:2813-2925)
def test_kat_issue14_tilted_beam_power(oracle):  # runtests.jl:2814-2829
    pd = bmo.Photodetector(10e-3, 1000)
    bmo.zrotate3d(pd, math.radians(90))
    bmo.translate3d(pd, [0.46, 0, 0])
    y_0 = 0.2
    beam = bmo.GaussianBeamlet([0, y_0, 0], [0.46, -y_0, 0], 532e-9, 2.5e-3, P0=10e-3)
    pd.empty()
    oracle.solve_system(bmo.System([pd]), beam)
    assert abs(pd.optical_power() - 10e-3) <= 1e-5


def _issue22_setup(n_substrate=1.5):
    splitter = bmo.CubeBeamsplitter(10 * mm, lambda lam: 1)
    substrate_length = 10 * mm
    substrate = bmo.Lens(bmo.shapes.CylinderSDF(5 * mm, substrate_length / 2), n_substrate)  # TestSubstrate <: AbstractRefractiveOptic
    detector = bmo.Photodetector(10 * mm, 250)
    bmo.translate3d(substrate, [0, -25 * mm, 0])
    bmo.translate3d(detector, [0, 40 * mm, 0])
    return bmo.System([substrate, splitter, detector]), substrate, detector, substrate_length


def _set_index(substrate, n):
    substrate.n = bmo.components._as_index(n)


def _shift_phase(gb, phase):  # electric_field!(gb, electric_field(gb) * exp(im*phase))
    gb.E0 = gb.E0 * np.exp(1j * phase)


def _ref_signal(phi, A):
    return (math.cos(phi) + 1) / 2 * A


def visibility(I):  # OpticUtils.jl interferometric visibility
    return (I.max() - I.min()) / (I.max() + I.min())


@pytest.mark.parametrize("index", [1, 10, 100, 1000])
def test_kat_issue22_visibility_for_any_substrate_index(oracle, index):  # runtests.jl:2866-2883
    system, substrate, detector, L = _issue22_setup()
    start_offset = 50 * mm
    _set_index(substrate, index)
    phis = bmo.linalg.linrange(0, 2 * math.pi, 30)
    power = np.zeros(len(phis))
    for i, p in enumerate(phis):
        gb_prb = bmo.GaussianBeamlet([0, -start_offset, 0], [0, 1, 0], 1e-6, 0.5 * mm)
        gb_ref = bmo.GaussianBeamlet([start_offset, 0, 0], [-1, 0, 0], 1e-6, 0.5 * mm)
        detector.empty()
        _shift_phase(gb_ref, p)
        oracle.solve_system(system, gb_prb)
        oracle.solve_system(system, gb_ref)
        power[i] = detector.optical_power()
    assert abs(visibility(power) - 1) <= 1e-2


def issue23_index_phase_shift(solve, n_steps):
    """runtests.jl:2885-2908: the substrate index rises until one more wavelength of OPL is in the probe arm; both beams are
    re-solved (retraced) every step."""
    system, substrate, detector, L = _issue22_setup()
    lam, start_offset = 1e-6, 50 * mm
    gb_prb = bmo.GaussianBeamlet([0, -start_offset, 0], [0, 1, 0], lam, 0.5 * mm)
    gb_ref = bmo.GaussianBeamlet([start_offset, 0, 0], [-1, 0, 0], lam, 0.5 * mm)
    n_lambdas = L / lam
    for nf in bmo.linalg.linrange(0, 1, n_steps):
        _set_index(substrate, 1 + 1 / n_lambdas * nf)
        detector.empty()
        solve(system, gb_prb)
        solve(system, gb_ref)
        assert abs(detector.optical_power() - _ref_signal(2 * math.pi * nf, 2e-3)) <= 1e-8
    delta = (gb_prb.optical_path_length() - gb_ref.optical_path_length()) / lam
    assert math.isclose(delta, 1, rel_tol=1.5e-8)
    return gb_prb, gb_ref


def test_kat_issue23_index_based_phase_shift(oracle):
    issue23_index_phase_shift(oracle.solve_system, 50)


def issue23_field_mutation(solve, n_steps):
    """runtests.jl:2910-2923: the probe beamlet's E0 is rotated between solves; the retrace must carry the new E0 into the stored
    children (_modify_beam_head!, Gaussian.jl:154-161).  The reference's testsets share the mutable substrate: the preceding
    testset leaves its index at 1 + lambda/L (one extra wavelength of OPL), which is what makes ref_signal(phi) exact to 1e-8
    here (with the constructor's n = 1.5 the plate adds a Gouy/curvature phase of 2.1e-3 rad between the arms)."""
    system, substrate, detector, L = _issue22_setup()
    _set_index(substrate, 1 + 1 / (L / 1e-6))
    start_offset = 50 * mm
    gb_prb = bmo.GaussianBeamlet([0, -start_offset, 0], [0, 1, 0], 1e-6, 0.5 * mm)
    gb_ref = bmo.GaussianBeamlet([start_offset, 0, 0], [-1, 0, 0], 1e-6, 0.5 * mm)
    phis = bmo.linalg.linrange(0, 2 * math.pi, n_steps)
    for phi in phis:
        detector.empty()
        solve(system, gb_prb)
        solve(system, gb_ref)
        assert abs(detector.optical_power() - _ref_signal(phi, 2e-3)) <= 1e-8
        _shift_phase(gb_prb, phis[1] - phis[0])
    return gb_prb, gb_ref


def test_kat_issue23_field_mutation_during_retracing(oracle):
    issue23_field_mutation(oracle.solve_system, 50)


@pytest.mark.gpu
def test_gpu_kat_issue23_on_the_engine():
    for b in issue23_index_phase_shift(bmo.solve_system, 12) + issue23_field_mutation(bmo.solve_system, 12):
        bmo.release(b)
