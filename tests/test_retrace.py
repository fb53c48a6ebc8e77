"""Retracing: the second solve_system! on solved beams — retrace_system! src/System.jl:188-255 (Beam),
:326-428 (GaussianBeamlet), solve_system! :444-461.

CPU part: the oracle reproduces the reference's retrace KATs (test/runtests.jl:2267-2287, :2568-2591, :2619-2651), and the
engine's lane code (host emulator) equals the oracle bit for bit on perturbed BASELINE scenes.
"""
import math

import numpy as np
import pytest

import bmo_amd as bmo
import scenes
from parity import compare, emu_trace

mm = 1e-3


def approx(a, b, atol=1e-12):
    return np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1.4901161193847656e-8, atol=atol)


# ------------------------------------------------------------------------------------------------ reference KATs
def test_kat_three_mirrors_retrace_after_move(oracle):  # runtests.jl:2243-2287
    m1, m2, m3 = (bmo.SquarePlanoMirror2D(1.0) for _ in range(3))
    bmo.translate3d(m2, [2, 0, 0])
    bmo.translate3d(m3, [2, 2, 0])
    bmo.zrotate3d(m1, math.radians(-90))
    bmo.yrotate3d(m1, math.radians(45))
    bmo.zrotate3d(m2, math.radians(45))
    bmo.xrotate3d(m3, math.radians(135))
    system = bmo.StaticSystem([m1, m2, m3])
    ray = bmo.PolarizedRay([0.0, 0, -2], [0, 0, 1], 1000e-9, [1, 0, 0])
    beam = bmo.Beam(ray)
    oracle.solve_system(system, beam)
    E = [r.E0 for r in beam.rays]
    assert approx(E[0].real, [1, 0, 0]) and approx(E[1].real, [0, 0, -1]) and approx(E[2].real, [0, 0, 1]) and approx(E[3].real, [0, -1, 0])
    assert beam.length() == 6.0
    # "y-Polarization": polarization!(ray, lin_y_pol); translate3d!(m3, [0, 2, 0]); solve_system! retraces
    beam.rays[0].E0 = np.array([0, 5, 0], dtype=np.complex128)
    bmo.translate3d(m3, [0, 2, 0])
    res = oracle.solve_system(system, beam)
    E = [r.E0 for r in beam.rays]
    assert approx(E[0].real, [0, 5, 0]) and approx(E[1].real, [0, -5, 0]) and approx(E[2].real, [5, 0, 0]) and approx(E[3].real, [-5, 0, 0])
    assert beam.length() == 8.0
    assert res.n_intersect_calls == 3 + 3  # three re-walked rays (one intersect3d each), then trace_all of the open tail


def test_kat_plate_beamsplitter_retrace_backside(oracle):  # runtests.jl:2540-2591
    N0 = 1.5
    pbs = bmo.RectangularPlateBeamsplitter(36 * mm, 25 * mm, 1 * mm, lambda n: N0)
    system = bmo.System([pbs])
    beam = bmo.Beam([0, -50 * mm, 0], [0, 1, 0], 1e-6)
    bmo.zrotate3d(pbs, math.radians(45))
    oracle.solve_system(system, beam)
    p, t, r = beam.rays, beam.children[0].rays, beam.children[1].rays
    assert (len(p), len(t), len(r)) == (1, 2, 1)
    bmo.zrotate3d(pbs, math.pi)
    oracle.solve_system(system, beam)  # retrace: the stored coating hit turns into a substrate hit, children are rebuilt
    p, t, r = beam.rays, beam.children[0].rays, beam.children[1].rays
    assert (len(p), len(t), len(r)) == (2, 1, 2)
    assert [x.n for x in p] == [1, N0] and [x.n for x in t] == [1] and [x.n for x in r] == [N0, 1]
    assert approx(p[0].dir, t[-1].dir) and approx(r[-1].dir, [1, 0, 0])


def test_kat_cube_beamsplitter_retrace_rotations(oracle):  # runtests.jl:2594-2651
    N0 = 1.5
    cbs = bmo.CubeBeamsplitter(25e-3, lambda n: N0)
    bmo.translate3d(cbs, [0, 50 * mm, 0])
    system = bmo.System([cbs])
    beam = bmo.Beam([0, 0, 0], [0, 1, 0], 1e-6)
    oracle.solve_system(system, beam)
    bmo.zrotate3d(cbs, math.pi / 2)
    oracle.solve_system(system, beam)
    t = beam.children[0].rays
    # :2629-2630 are exact in the reference; one ulp of cos(pi/2) here — see tests/test_oracle_kat2.py::test_cube_beamsplitter_rotated_90
    assert np.abs(t[-1].dir - beam.rays[0].dir).max() <= 6.2e-17 and np.abs(t[-1].dir - np.array([0, 1, 0])).max() <= 6.2e-17
    bmo.zrotate3d(cbs, math.pi / 2)
    oracle.solve_system(system, beam)
    p, t, r = beam.rays, beam.children[0].rays, beam.children[1].rays
    assert (len(p), len(t), len(r)) == (2, 2, 2)
    assert [x.n for x in p] == [1, N0] and [x.n for x in t] == [N0, 1] and [x.n for x in r] == [N0, 1]
    assert approx(t[-1].dir, p[0].dir) and approx(r[-1].dir, [-1, 0, 0])


def test_retrace_of_unchanged_system_reproduces_the_solution(oracle):
    """Re-walking an unmoved system gives the same rays; only the call count differs (1 per stored ray instead of M)."""
    system, _ = scenes.c2_scene()
    bundle = scenes.c2_bundle(64)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    a, sol = oracle.trace(scene, bundle, 20, keep=True)
    b = oracle.trace(scene, bundle, 20, prev=sol)
    b.n_intersect_calls = a.n_intersect_calls
    compare(b, a, rtol=0.0, label="retrace(unchanged)")


# ------------------------------------------------------------------------------------------------ lane code == oracle
def _perturb_c2(parts, case):
    bs, det_t, det_r = parts["bs"], parts["det_t"], parts["det_r"]
    if case == "tilt_splitter":
        bmo.xrotate3d(bs, math.radians(0.7))
    elif case == "move_detector_away":  # the transmitted arm now runs into nothing: stored last rays lose their intersection
        bmo.translate3d(det_t, [0, 0.5, 0])
    elif case == "shift_train":  # a lens moves sideways: some stored hits are lost -> cut, children dropped, fresh trace
        bmo.translate3d(parts["objs"][0], [0.9 * mm, 0, 0])
    elif case == "remove_splitter_from_path":
        bmo.translate3d(bs, [0.2, 0, 0])
    else:
        raise ValueError(case)


def _c2_with_parts():
    objs = scenes.miniscope_objects()
    y0 = 0.332 * mm
    bs = bmo.ThinBeamsplitter(10 * mm)
    bmo.xrotate3d(bs, math.radians(45))
    bmo.translate3d(bs, [0, y0, 30 * mm])
    det_t = bmo.Spotdetector(5 * mm)
    bmo.xrotate3d(det_t, math.radians(90))
    bmo.translate3d(det_t, [0, y0, 34 * mm])
    det_r = bmo.Spotdetector(5 * mm)
    bmo.translate3d(det_r, [0, y0 - 4 * mm, 30 * mm])
    return bmo.System(objs + [bs, det_t, det_r]), dict(objs=objs, bs=bs, det_t=det_t, det_r=det_r)


RETRACE_CASES = ["tilt_splitter", "move_detector_away", "shift_train", "remove_splitter_from_path"]


def retrace_pair(kind, case, n, r_max=20):
    """(scene0, scene1, bundle): a BASELINE scene before and after a kinematic move."""
    system, parts = _c2_with_parts()
    if kind == "ray":
        bundle = scenes.c2_bundle(n)
    elif kind == "gauss":
        bundle = scenes.c3_bundle(n)
    else:
        bundle = scenes.polarized_bundle(n, center=[0, -5 * mm, 0], direction=[0, 1, 0], diameter=1.2 * mm)
    scene0 = bmo.CompiledScene(system, bundle.lambdas)
    _perturb_c2(parts, case)
    scene1 = bmo.CompiledScene(system, bundle.lambdas)
    return scene0, scene1, bundle


@pytest.mark.parametrize("case", RETRACE_CASES)
@pytest.mark.parametrize("kind", ["ray", "gauss", "pol"])
def test_emulator_retrace_equals_oracle(oracle, kind, case):
    scene0, scene1, bundle = retrace_pair(kind, case, 96)
    a0, sol = oracle.trace(scene0, bundle, 20, keep=True)
    a1, sol1 = oracle.trace(scene1, bundle, 20, keep=True, prev=sol)
    e0 = emu_trace(scene0, bundle, 20)
    compare(e0, a0, rtol=0.0, label=f"{kind}/{case} first solve")
    e1 = emu_trace(scene1, bundle, 20, prev=e0)
    assert not (a1.node_status & 512).any()
    compare(e1, a1, rtol=0.0, label=f"{kind}/{case} retrace")
    # the move really changed the solution, and the retrace is not simply a fresh trace of the moved system
    fresh = oracle.trace(scene1, bundle, 20)
    assert a1.n_intersect_calls != fresh.n_intersect_calls
    # a third solve (retrace of a retraced solution) after moving back
    a2 = oracle.trace(scene0, bundle, 20, prev=sol1)
    e2 = emu_trace(scene0, bundle, 20, prev=e1)
    compare(e2, a2, rtol=0.0, label=f"{kind}/{case} retrace back")


def test_retrace_with_smaller_r_max(oracle):
    scene0, scene1, bundle = retrace_pair("ray", "shift_train", 64)
    a0, sol = oracle.trace(scene0, bundle, 20, keep=True)
    e0 = emu_trace(scene0, bundle, 20)
    for r_max in (1, 2, 3, 5, 8):
        a1 = oracle.trace(scene1, bundle, r_max, prev=sol)
        e1 = emu_trace(scene1, bundle, r_max, prev=e0)
        compare(e1, a1, rtol=0.0, label=f"r_max={r_max}")


def test_retrace_rejects_mismatched_batch(oracle):
    scene0, scene1, bundle = retrace_pair("ray", "tilt_splitter", 16)
    a0, sol = oracle.trace(scene0, bundle, 20, keep=True)
    with pytest.raises(RuntimeError):
        oracle.trace(scene1, scenes.c2_bundle(8), 20, prev=sol)


# ------------------------------------------------------------------------------------------------ HIP engine == oracle
def _engine_solve(scene, bundle, r_max, prev=None):
    return bmo.system._engine_solve(scene, bundle, r_max, prev)


@pytest.mark.gpu
@pytest.mark.parametrize("case", RETRACE_CASES)
@pytest.mark.parametrize("kind,rtol", [("ray", 0.0), ("gauss", 0.0), ("pol", 0.0)])
def test_gpu_retrace_equals_oracle(oracle, kind, rtol, case):
    scene0, scene1, bundle = retrace_pair(kind, case, 2048)
    a0, sol = oracle.trace(scene0, bundle, 20, threads=16, keep=True)
    a1, sol1 = oracle.trace(scene1, bundle, 20, threads=16, keep=True, prev=sol)
    a2 = oracle.trace(scene0, bundle, 20, threads=16, prev=sol1)
    g0, gs0 = _engine_solve(scene0, bundle, 20)
    compare(g0, a0, rtol, f"{kind}/{case} first solve")
    g1, gs1 = _engine_solve(scene1, bundle, 20, gs0)
    compare(g1, a1, rtol, f"{kind}/{case} retrace")
    g1b, gs1b = _engine_solve(scene1, bundle, 20, gs0)  # the previous solution is not consumed
    compare(g1b, g1, 0.0, f"{kind}/{case} retrace twice from the same solution")
    g2, gs2 = _engine_solve(scene0, bundle, 20, gs1)
    compare(g2, a2, rtol, f"{kind}/{case} retrace back")
    for s in (gs0, gs1, gs1b, gs2):
        s.free()


@pytest.mark.gpu
@pytest.mark.parametrize("r_max", [1, 2, 3, 8])
def test_gpu_retrace_r_max(oracle, r_max):
    scene0, scene1, bundle = retrace_pair("ray", "shift_train", 512)
    a0, sol = oracle.trace(scene0, bundle, 20, threads=16, keep=True)
    a1 = oracle.trace(scene1, bundle, r_max, threads=16, prev=sol)
    g0, gs0 = _engine_solve(scene0, bundle, 20)
    g1, gs1 = _engine_solve(scene1, bundle, r_max, gs0)
    compare(g1, a1, 0.0, f"r_max={r_max}")


@pytest.mark.gpu
def test_gpu_retrace_kats_through_the_host_mirror():
    """The reference's plate-splitter retrace KAT (runtests.jl:2568-2591) through solve_system on the engine."""
    N0 = 1.5
    pbs = bmo.RectangularPlateBeamsplitter(36 * mm, 25 * mm, 1 * mm, lambda n: N0)
    system = bmo.System([pbs])
    beam = bmo.Beam([0, -50 * mm, 0], [0, 1, 0], 1e-6)
    bmo.zrotate3d(pbs, math.radians(45))
    bmo.solve_system(system, beam)
    assert (len(beam.rays), len(beam.children[0].rays), len(beam.children[1].rays)) == (1, 2, 1)
    bmo.zrotate3d(pbs, math.pi)
    bmo.solve_system(system, beam)
    p, t, r = beam.rays, beam.children[0].rays, beam.children[1].rays
    assert (len(p), len(t), len(r)) == (2, 1, 2)
    assert [x.n for x in p] == [1, N0] and [x.n for x in t] == [1] and [x.n for x in r] == [N0, 1]
    assert approx(p[0].dir, t[-1].dir) and approx(r[-1].dir, [1, 0, 0])
    bmo.release(beam)


@pytest.mark.gpu
def test_gpu_retrace_rejects_mismatched_batch():
    scene0, scene1, bundle = retrace_pair("ray", "tilt_splitter", 64)
    g0, gs0 = _engine_solve(scene0, bundle, 20)
    with pytest.raises(RuntimeError):
        _engine_solve(scene1, scenes.c2_bundle(32), 20, gs0)


@pytest.mark.gpu
def test_solve_again_without_retrace_traces_only_open_leaves():
    """solve_system!(...; retrace = false) on solved beams (System.jl:449-458, :470-475): beams whose last ray has an intersection are
    left alone (their detector gets no second record), beams whose last ray is open are traced on from that ray — with the OPL of
    the whole beam in a PSF record, exactly as if the new object had been there in the first solve."""
    import math

    from bmo_amd import components as cp

    def build(with_psf):
        lens = bmo.SphericalLens(60 * mm, -60 * mm, 5 * mm, 25.4 * mm, 1.5)
        spot = bmo.Spotdetector(4 * mm)            # small: catches the inner rays only
        bmo.translate3d(spot, [0, 30 * mm, 0])
        objs = [lens, spot]
        psf = None
        if with_psf:
            psf = bmo.PSFDetector(60 * mm)
            bmo.translate3d(psf, [0, 50 * mm, 0])
            objs.append(psf)
        return bmo.System(objs), spot, psf

    def source():
        return bmo.UniformDiscSource([0, -20 * mm, 0], [0, 1, 0], 12 * mm, 1.064e-6, num_rays=200, e1=[1, 0, 0])

    # reference behaviour of a fresh solve with the PSF detector in place from the start
    sys_full, spot_full, psf_full = build(True)
    fresh = source()
    bmo.solve_system(sys_full, fresh)
    # first solve without it, then the detector appears and the beams are solved again without retracing
    sys0, spot0, _ = build(False)
    again = source()
    bmo.solve_system(sys0, again)
    n_spot = len(spot0.data)
    open_before = sum(1 for b in again.beams if b.rays[-1].intersection is None)
    assert 0 < n_spot < 200 and open_before == 200 - n_spot
    sys1 = bmo.System(sys0.objects() + [bmo.PSFDetector(60 * mm)])
    psf1 = sys1.objects()[-1]
    bmo.translate3d(psf1, [0, 50 * mm, 0])
    assert bmo.solve_system(sys1, again, retrace=False) is None
    assert len(spot0.data) == n_spot                                   # closed beams were not traced again
    assert len(psf1.data) == open_before == len(psf_full.data)         # open ones went on to the new detector
    assert np.array_equal(psf1.data, psf_full.data)                    # hit, dir, OPL of the WHOLE beam, proj, k: as in the fresh solve
    for a, b in zip(again.beams, fresh.beams):
        assert len(a.rays) == len(b.rays)
        for ra, rb in zip(a.rays, b.rays):
            assert np.array_equal(ra.pos, rb.pos) and np.array_equal(ra.dir, rb.dir)
            assert (ra.intersection is None) == (rb.intersection is None)
            if ra.intersection is not None:
                assert ra.intersection.t == rb.intersection.t and type(ra.intersection.object) is type(rb.intersection.object)
    bmo.release(again)
    bmo.release(fresh)


@pytest.mark.gpu
def test_solve_again_without_retrace_continues_open_gaussian_beamlets():
    """The same for GaussianBeamlets: the open beamlets (here: both children of a first splitter) are traced on as a batch that brings
    their accumulated lengths along (bmo.h "31 planes": lenA, lenB, l0 and the three optical paths), so a SECOND splitter further on
    sizes its children (w0, E0: gauss_parameters at the whole beamlet's length, ThinBeamsplitter.jl:117-168) and a PSFDetector records
    the optical path of the whole beamlet exactly as in a solve that never stopped."""
    import math

    def build(stage):
        bs1 = bmo.ThinBeamsplitter(20 * mm)
        bmo.xrotate3d(bs1, math.radians(45))
        bmo.translate3d(bs1, [0, 40 * mm, 0])
        objs = [bs1]
        if stage >= 1:
            lens = bmo.SphericalLens(80 * mm, -80 * mm, 4 * mm, 25.4 * mm, 1.5)
            bmo.translate3d(lens, [0, 90 * mm, 0])
            bs2 = bmo.ThinBeamsplitter(20 * mm)
            bmo.xrotate3d(bs2, math.radians(45))
            bmo.translate3d(bs2, [0, 140 * mm, 0])
            psf = bmo.PSFDetector(30 * mm)
            bmo.translate3d(psf, [0, 200 * mm, 0])
            objs += [lens, bs2, psf]
        return bmo.System(objs)

    def beamlets():
        return [bmo.GaussianBeamlet([0.2 * mm * i, 0, 0.1 * mm * i], [0.0, 1, 0], 1064e-9, 0.4e-3, support=[1, 0, 0]) for i in range(-2, 3)]

    fresh = beamlets()
    sys_full = build(1)
    for g in fresh:
        bmo.solve_system(sys_full, g)
    again = beamlets()
    sys0 = build(0)
    for g in again:
        bmo.solve_system(sys0, g)
        assert len(g.children) == 2 and all(c.rays[-1].intersection is None for c in g.children)  # both arms open
    sys1 = build(1)
    for g in again:
        assert bmo.solve_system(sys1, g, retrace=False) is None

    def same(a, b, path="root"):
        assert a.w0 == b.w0 and a.E0 == b.E0, path
        for pa, pb in ((a.chief, b.chief), (a.waist, b.waist), (a.divergence, b.divergence)):
            assert len(pa.rays) == len(pb.rays), path
            for ra, rb in zip(pa.rays, pb.rays):
                assert np.array_equal(ra.pos, rb.pos) and np.array_equal(ra.dir, rb.dir) and ra.n == rb.n, path
                assert (ra.intersection is None) == (rb.intersection is None), path
                if ra.intersection is not None:
                    assert ra.intersection.t == rb.intersection.t and np.array_equal(ra.intersection.n, rb.intersection.n), path
        assert len(a.children) == len(b.children), path
        for q, (ca, cb) in enumerate(zip(a.children, b.children)):
            assert ca.parent is a and ca.chief.parent is a.chief
            same(ca, cb, path + ".%d" % q)

    deep = 0
    for a, b in zip(again, fresh):
        same(a, b)
        deep += sum(len(c.children) for c in a.children)
        assert a.length() == b.length() and all(ca.length() == cb.length() for ca, cb in zip(a.children, b.children))
    assert deep > 0  # the second splitter was reached in the continuation and sized its children from the seeded lengths
    psf_a, psf_b = sys1.objects()[-1], sys_full.objects()[-1]
    assert len(psf_a.data) == len(psf_b.data) > 0 and np.array_equal(np.asarray(psf_a.data), np.asarray(psf_b.data))
    for x in again + fresh:
        bmo.release(x)


@pytest.mark.gpu
def test_continued_beamlets_reach_a_photodetector():
    """solve_system!(...; retrace = false) on solved GaussianBeamlets whose continuation ends on a Photodetector (Photodetector.jl:69-107):
    the field is a function of every ray of the beamlet (point_on_beam / length / optical_path_length, Beam.jl:125-205), the rays in front of
    the open one go to the device as a prefix (bmo_result_set_gauss_prefix).  Same field as a solve that never stopped."""
    mm = 1e-3

    def build():
        l1 = bmo.SphericalLens(50 * mm, -50 * mm, 5 * mm, 20 * mm, 1.5)
        bmo.translate3d(l1, [0, 20 * mm, 0])
        l2 = bmo.SphericalLens(80 * mm, -80 * mm, 4 * mm, 20 * mm, 1.6)
        bmo.translate3d(l2, [0, 60 * mm, 0])
        pd = bmo.Photodetector(4 * mm, 24)
        bmo.translate3d(pd, [0, 140 * mm, 0])
        return bmo.System([l1, l2, pd]), pd

    def beamlets():
        return [bmo.GaussianBeamlet([dx * mm, 0, 0.1 * dx * mm], [0.01 * dx, 1.0, 0.0], 1064e-9, 0.4 * mm, support=[1, 0, 0]) for dx in (-1.0, 0.0, 0.7, 1.3)]

    sys_a, pd_a = build()
    stopped = beamlets()
    for g in stopped:
        bmo.solve_system(sys_a, g, r_max=3)  # stops inside the second lens: three rays, the last one open
        assert len(g.chief.rays) == 3 and g.chief.rays[-1].intersection is None
    assert not np.any(pd_a.field)
    for g in stopped:
        assert bmo.solve_system(sys_a, g, r_max=20, retrace=False) is None
        assert g.chief.rays[-1].intersection is not None and len(g.chief.rays) == 5
    sys_b, pd_b = build()
    full = beamlets()
    for g in full:
        bmo.solve_system(sys_b, g, r_max=20)
    assert np.any(pd_b.field)
    np.testing.assert_allclose(pd_a.field, pd_b.field, rtol=1e-12, atol=1e-12 * np.abs(pd_b.field).max())
    for x in stopped + full:
        bmo.release(x)


@pytest.mark.gpu
def test_children_born_in_a_continuation_get_the_full_r_max():
    """solve_system!(...; retrace = false): the continued beam has r_max - length rays left, the children a splitter makes in the continuation
    have all of r_max (every Beam counts its own rays, System.jl:118-154) — and detector rows come in the reference's order, root by root."""
    mm = 1e-3

    def build(stage):
        l1 = bmo.SphericalLens(60 * mm, -60 * mm, 4 * mm, 20 * mm, 1.5)
        bmo.translate3d(l1, [0, 20 * mm, 0])
        objs = [l1]
        if stage >= 1:
            bs = bmo.ThinBeamsplitter(30 * mm)
            bmo.xrotate3d(bs, math.radians(45))
            bmo.translate3d(bs, [0, 60 * mm, 0])
            l2 = bmo.SphericalLens(70 * mm, -70 * mm, 4 * mm, 20 * mm, 1.5)
            bmo.translate3d(l2, [0, 90 * mm, 0])
            l3 = bmo.SphericalLens(90 * mm, -90 * mm, 3 * mm, 20 * mm, 1.6)
            bmo.translate3d(l3, [0, 110 * mm, 0])
            sd = bmo.Spotdetector(20 * mm)
            bmo.translate3d(sd, [0, 150 * mm, 0])
            objs += [bs, l2, l3, sd]
        return bmo.System(objs)

    def beams():
        return [bmo.Beam(bmo.Ray([dx * mm, 0, 0.3 * dx * mm], [0, 1.0, 0], 1e-6)) for dx in (0.0, 1.0, -2.0, 3.0)]

    r_max = 6
    sys_full = build(1)
    fresh = beams()
    for b in fresh:
        bmo.solve_system(sys_full, b, r_max=r_max)
    sd_full = sys_full.objects()[-1]
    assert len(sd_full.data) == len(fresh)  # the transmitted child takes 5 rays to the detector: more than the 4 the continued beam has left
    again = beams()
    sys0 = build(0)
    for b in again:
        bmo.solve_system(sys0, b, r_max=r_max)
        assert len(b.rays) == 3 and b.rays[-1].intersection is None
    sys1 = build(1)
    for b in again:
        assert bmo.solve_system(sys1, b, r_max=r_max, retrace=False) is None
    sd1 = sys1.objects()[-1]
    assert np.array_equal(np.asarray(sd1.data), np.asarray(sd_full.data))

    def same(a, b):
        assert len(a.rays) == len(b.rays) and len(a.children) == len(b.children)
        for ra, rb in zip(a.rays, b.rays):
            assert np.array_equal(ra.pos, rb.pos) and np.array_equal(ra.dir, rb.dir)
            assert (ra.intersection is None) == (rb.intersection is None)
            if ra.intersection is not None:
                assert ra.intersection.t == rb.intersection.t
        for ca, cb in zip(a.children, b.children):
            same(ca, cb)

    for a, b in zip(again, fresh):
        same(a, b)
    for x in again + fresh:
        bmo.release(x)


@pytest.mark.gpu
def test_continuation_of_a_beam_list_flushes_detector_rows_root_by_root():
    """solve_system!(system, [b1, b2, ...]; retrace = false): two splitters in a row send several beams of every tree to ONE
    detector at different depths; the reference's loop finishes root 1's whole tree before root 2 (System.jl:463-468), so the rows of
    root 1 — all depths — come before any row of root 2.  Compared with a solve of the same list that never stopped."""
    mm = 1e-3

    def build(stage):
        l1 = bmo.SphericalLens(60 * mm, -60 * mm, 4 * mm, 20 * mm, 1.5)
        bmo.translate3d(l1, [0, 20 * mm, 0])
        objs = [l1]
        if stage >= 1:
            bs1 = bmo.ThinBeamsplitter(30 * mm)
            bmo.xrotate3d(bs1, math.radians(45))
            bmo.translate3d(bs1, [0, 60 * mm, 0])
            bs2 = bmo.ThinBeamsplitter(30 * mm)
            bmo.xrotate3d(bs2, math.radians(45))
            bmo.translate3d(bs2, [0, 90 * mm, 0])
            # bs1 reflects +y into -z; this mirror folds that arm back onto +y (past bs2) so that it meets the detector too, one tree
            # level above bs2's transmitted arm
            m = bmo.SquarePlanoMirror2D(30 * mm)
            bmo.xrotate3d(m, math.radians(45))
            bmo.translate3d(m, [0, 60 * mm, -30 * mm])
            sd = bmo.Spotdetector(120 * mm)
            bmo.translate3d(sd, [0, 150 * mm, 10 * mm])
            objs += [bs1, bs2, m, sd]
        return bmo.System(objs)

    def beams():
        return [bmo.Beam(bmo.Ray([dx * mm, 0, 0.3 * dx * mm], [0, 1.0, 0], 1e-6)) for dx in (0.0, 1.0, -2.0)]

    r_max = 10
    sys_full = build(1)
    fresh = beams()
    bmo.solve_system(sys_full, fresh, r_max=r_max)
    sd_full = sys_full.objects()[-1]
    assert len(sd_full.data) >= 2 * len(fresh)  # at least two beams of every tree end on the detector
    again = beams()
    bmo.solve_system(build(0), again, r_max=r_max)
    assert all(b.rays[-1].intersection is None for b in again)
    sys1 = build(1)
    assert bmo.solve_system(sys1, again, r_max=r_max, retrace=False) is None
    assert np.array_equal(np.asarray(sys1.objects()[-1].data), np.asarray(sd_full.data))
    for x in (again, fresh):
        bmo.release(x)


@pytest.mark.gpu
def test_gauss_prefix_argument_checks():
    """bmo_result_set_gauss_prefix refuses what it cannot mean: a Ray solution, a wrong root count, starts that do not begin at 0 or decrease."""
    import ctypes as C

    mm = 1e-3
    lens = bmo.SphericalLens(60 * mm, -60 * mm, 4 * mm, 20 * mm, 1.5)
    bmo.translate3d(lens, [0, 20 * mm, 0])
    system = bmo.System([lens])
    g = [bmo.GaussianBeamlet([0.1 * mm * i, 0, 0], [0, 1.0, 0], 1064e-9, 0.4 * mm, support=[1, 0, 0]) for i in range(3)]
    for b in g:
        bmo.solve_system(system, b)
    sol = g[0]._solution
    lib = sol.lib
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)

    def call(n_roots, starts, segs, opl):
        st = np.ascontiguousarray(starts, dtype=np.int32)
        sg = np.ascontiguousarray(segs, dtype=np.float64)
        op = np.ascontiguousarray(opl, dtype=np.float64)
        return lib.bmo_result_set_gauss_prefix(sol.handle, n_roots, st.ctypes.data_as(ip), sg.ctypes.data_as(dp), op.ctypes.data_as(dp))

    assert call(1, [0, 0], np.zeros((24, 0)), [0.0]) == 0                      # no earlier segments: fine
    assert call(1, [0, 2], np.ones((24, 2)), [0.5]) == 0                       # replaces the prefix
    assert call(2, [0, 0, 0], np.zeros((24, 0)), [0.0, 0.0]) != 0              # the solution has one root
    assert call(1, [1, 1], np.zeros((24, 1)), [0.0]) != 0                      # starts begin at 0
    assert call(1, [0, -1], np.zeros((24, 0)), [0.0]) != 0                     # ... and do not decrease
    ray = bmo.Beam(bmo.Ray([0, 0, 0], [0, 1.0, 0], 1e-6))
    bmo.solve_system(system, ray)
    rsol = ray._solution
    st = np.zeros(2, dtype=np.int32)
    z = np.zeros(1)
    assert rsol.lib.bmo_result_set_gauss_prefix(rsol.handle, 1, st.ctypes.data_as(ip), z.ctypes.data_as(dp), z.ctypes.data_as(dp)) != 0  # not a beamlet solution
    for x in g + [ray]:
        bmo.release(x)
