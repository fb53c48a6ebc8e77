"""Randomised parity: seeded random element trains (spherical / meniscus / doublet / aspheric / cylinder lenses, prisms, plate, cube and
thin beam splitters, mirrors, detectors; random tilts and decentres) and random bundles of all three beam kinds.

CPU: the engine's per-lane code (host emulator build of csrc/bmo_lane.hpp) must reproduce the oracle bit-exactly — this is what
keeps the engine's result-preserving shortcuts (bounding-sphere culls, nearest-hit prune, union child skipping, hint reuse, start
classification probe) honest on geometry nobody hand-picked.   GPU: the same scenes through the C ABI.
Bar as everywhere: ids / counts / tree bit-exact, FP64 planes bit-exact for every beam kind (round 4: no C library enters a trace,
tests/test_jl_trig.py).
"""
import math

import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import disc_bundle, gaussian_bundle, polarized_bundle

mm = 1e-3
D = 25.4 * mm
R_MAX = 30


def _radius(rng, allow_inf=True):
    if allow_inf and rng.random() < 0.2:
        return math.inf
    return float(rng.choice([-1.0, 1.0]) * rng.uniform(25, 120) * mm)


def _element(rng, allow_splitter):
    """One random element centred near the origin with its optical axis along +y; returns (object, axial length, is_splitter)."""
    kind = rng.choice(["singlet", "singlet", "doublet", "asphere", "asphere2", "cylinder", "acylinder", "plate_bs", "cube_bs", "thin_bs", "prism", "window"])
    if kind.endswith("_bs") and not allow_splitter:
        kind = "singlet"
    obj, length = _make(rng, kind)
    return obj, length, kind.endswith("_bs")


def _make(rng, kind):
    n = float(rng.uniform(1.4, 1.8))
    if kind == "singlet":
        while True:
            r1, r2 = _radius(rng), _radius(rng)
            if math.isinf(r1) and math.isinf(r2):
                r2 = -60 * mm
            l = float(rng.uniform(4, 8) * mm)
            try:
                return bmo.Lens(bmo.SphericalSurface(r1, D), bmo.SphericalSurface(r2, D), l, lambda lam, n=n: n), l
            except ValueError:  # radii / thickness without an edge (the builder refuses like the reference's): draw again
                continue
    if kind == "doublet":
        r1, r2, r3 = float(rng.uniform(40, 90) * mm), -float(rng.uniform(30, 60) * mm), -float(rng.uniform(80, 200) * mm)
        l1, l2 = float(rng.uniform(5, 8) * mm), float(rng.uniform(2, 3) * mm)
        return bmo.SphericalDoubletLens(r1, r2, r3, l1, l2, D, n, float(rng.uniform(1.5, 1.8))), l1 + l2
    if kind == "asphere":
        while True:
            l = float(rng.uniform(5, 7) * mm)
            r = float(rng.uniform(25, 60) * mm)
            front = bmo.EvenAsphericalSurface(r, D, float(rng.uniform(-1.2, 0.2)), [0.0, float(rng.uniform(-2, 2)), float(rng.uniform(-500, 500))])
            back = bmo.SphericalSurface(_radius(rng), D)
            try:
                return bmo.Lens(front, back, l, lambda lam, n=n: n), l
            except ValueError:
                continue
    if kind == "asphere2":  # concave or convex even asphere in front, even asphere behind
        while True:
            l = float(rng.uniform(5, 7) * mm)
            r1 = float(rng.choice([-1.0, 1.0]) * rng.uniform(30, 70) * mm)
            r2 = float(rng.choice([-1.0, 1.0]) * rng.uniform(30, 70) * mm)
            front = bmo.EvenAsphericalSurface(r1, D, float(rng.uniform(-1.5, 0.5)), [0.0, float(rng.uniform(-1, 1)), float(rng.uniform(-300, 300))])
            back = bmo.EvenAsphericalSurface(r2, D, float(rng.uniform(-1.5, 0.5)), [0.0, float(rng.uniform(-1, 1)), float(rng.uniform(-300, 300))])
            try:
                return bmo.Lens(front, back, l, lambda lam, n=n: n), l
            except ValueError:
                continue
    if kind == "acylinder":
        l = float(rng.uniform(5, 8) * mm)
        r = float(rng.choice([-1.0, 1.0]) * rng.uniform(14, 40) * mm)
        coeffs = [0.0, float(rng.uniform(-20, 20)), float(rng.uniform(-3e3, 3e3))]
        return bmo.Lens(bmo.AcylindricalSurface(r, 22 * mm, 24 * mm, float(rng.uniform(-1.2, 0.0)), coeffs), l, lambda lam, n=n: n), l
    if kind == "cylinder":
        l = float(rng.uniform(4, 6) * mm)
        r = float(rng.choice([-1.0, 1.0]) * rng.uniform(30, 80) * mm)
        return bmo.Lens(bmo.CylindricalSurface(r, 20 * mm, 22 * mm), l, lambda lam, n=n: n), l
    if kind == "plate_bs":
        o = bmo.RectangularPlateBeamsplitter(30 * mm, 30 * mm, 4 * mm, lambda lam, n=n: n, reflectance=float(rng.uniform(0.3, 0.7)))
        bmo.xrotate3d(o, math.radians(float(rng.uniform(20, 45))))
        return o, 25 * mm
    if kind == "cube_bs":
        o = bmo.CubeBeamsplitter(20 * mm, lambda lam, n=n: n)
        bmo.zrotate3d(o, math.radians(float(rng.uniform(-4, 4))))
        return o, 24 * mm
    if kind == "thin_bs":
        o = bmo.ThinBeamsplitter(30 * mm, reflectance=float(rng.uniform(0.3, 0.7)))
        bmo.xrotate3d(o, math.radians(float(rng.uniform(30, 50))))
        return o, 25 * mm
    if kind == "prism":
        o = bmo.RightAnglePrism(20 * mm, 25 * mm, lambda lam, n=n: n)
        bmo.zrotate3d(o, math.radians(float(rng.uniform(0, 360))))
        return o, 30 * mm
    l = float(rng.uniform(1, 3) * mm)
    return bmo.Lens(bmo.CircularFlatSurface(D), l, lambda lam, n=n: n), l


def random_system(seed, with_detectors=True):
    rng = np.random.Generator(np.random.PCG64(seed))
    objs = []
    y = 0.0
    n_bs = 0
    for _ in range(int(rng.integers(3, 7))):
        o, length, is_bs = _element(rng, n_bs == 0)  # one splitter per train: two facing each other multiply the beam tree without bound
        n_bs += is_bs
        bmo.xrotate3d(o, math.radians(float(rng.uniform(-4, 4))))
        bmo.zrotate3d(o, math.radians(float(rng.uniform(-4, 4))))
        bmo.translate3d(o, [float(rng.uniform(-1, 1) * mm), y, float(rng.uniform(-1, 1) * mm)])
        objs.append(o)
        y += length + float(rng.uniform(2, 15) * mm)
    end = rng.choice(["mirror", "concave_mirror", "retro", "spot", "psf", "none"])
    if end in ("mirror", "concave_mirror", "retro"):  # sends the bundle back through the train
        if end == "mirror":
            m = bmo.RoundPlanoMirror(30 * mm, 5 * mm)
        elif end == "concave_mirror":
            m = bmo.ConcaveSphericalMirror(float(rng.uniform(80, 300) * mm), 6 * mm, 30 * mm)
        else:
            m = bmo.Retroreflector(25 * mm)
        bmo.zrotate3d(m, math.radians(180 + float(rng.uniform(-2, 2))))
        bmo.xrotate3d(m, math.radians(float(rng.uniform(-2, 2))))
        bmo.translate3d(m, [0, y + 10 * mm, 0])
        objs.append(m)
    if not with_detectors and end not in ("mirror", "concave_mirror", "retro"):  # Gaussian bundles: a tilted Photodetector behind the train (f2)
        pd = bmo.Photodetector(30 * mm, 24)
        bmo.xrotate3d(pd, math.radians(float(rng.uniform(-5, 5))))
        bmo.zrotate3d(pd, math.radians(float(rng.uniform(-5, 5))))
        bmo.translate3d(pd, [0, y + 10 * mm, 0])
        objs.append(pd)
    if with_detectors:
        if end in ("spot", "psf"):
            d = bmo.Spotdetector(40 * mm) if end == "spot" else bmo.PSFDetector(40 * mm)
            bmo.xrotate3d(d, math.radians(float(rng.uniform(-5, 5))))
            bmo.translate3d(d, [0, y + 10 * mm, 0])
            objs.append(d)
        side = bmo.Spotdetector(60 * mm)  # catches reflected arms of splitters
        bmo.xrotate3d(side, math.radians(90))
        bmo.translate3d(side, [0, y / 2, -60 * mm])
        objs.append(side)
    return bmo.System(objs), rng


def random_bundle(rng, kind, n):
    center = [float(rng.uniform(-1, 1) * mm), -20 * mm, float(rng.uniform(-1, 1) * mm)]
    direction = [float(rng.uniform(-0.02, 0.02)), 1.0, float(rng.uniform(-0.02, 0.02))]
    seed = int(rng.integers(1, 1 << 30))
    lam = float(rng.choice([532e-9, 1064e-9]))
    if kind == "ray":
        return disc_bundle(n, center, direction, 0.7 * D, lam=lam, seed=seed, cone=float(rng.uniform(0.0, 0.06)))
    if kind == "pol":
        return polarized_bundle(n, center, direction, 0.6 * D, lam=lam, jitter=0.03, seed=seed)
    return gaussian_bundle(n, center, direction, 0.4 * D, lam=lam, w0=float(rng.uniform(30e-6, 200e-6)), seed=seed)


def _limit(n):
    return 400 * n  # beams: a draw whose splitter ends up facing a mirror multiplies beams without bound (neither side would finish)


def _engine_first(fn):
    """Run the engine solve `fn` (with max_beams set); skip the case when the beam tree runs away."""
    try:
        return fn()
    except RuntimeError as e:
        if "(-6)" in str(e):
            pytest.skip("runaway beam tree in this draw (BMO_ERR_LIMIT)")
        raise


CASES = [(seed, "ray") for seed in range(101, 165)] + [(seed, "pol") for seed in range(201, 217)] + [(seed, "gauss") for seed in range(301, 317)]


def _tol(kind):
    return 0.0


def _case(seed, kind, n):
    system, rng = random_system(seed, with_detectors=(kind != "gauss"))
    bundle = random_bundle(rng, kind, n)
    return bmo.CompiledScene(system, bundle.lambdas), bundle


@pytest.mark.parametrize("seed,kind", CASES)
def test_lane_code_equals_oracle_on_random_scenes(oracle, seed, kind):
    scene, bundle = _case(seed, kind, 128 if kind == "ray" else 64)
    got = _engine_first(lambda: emu_trace(scene, bundle, R_MAX, max_beams=_limit(bundle.n)))  # the lane code first: it can stop a runaway tree
    ref = oracle.trace(scene, bundle, R_MAX, threads=4)
    compare(got, ref, _tol(kind), "fuzz %d %s" % (seed, kind))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kind", CASES)
def test_engine_equals_oracle_on_random_scenes(oracle, seed, kind):
    scene, bundle = _case(seed, kind, 4096 if kind == "ray" else 1024)
    eng = bmo.Engine(scene, 0, max_beams=_limit(bundle.n))
    try:
        got = _engine_first(lambda: eng.trace(bundle, R_MAX))
    finally:
        eng.close()
    ref = oracle.trace(scene, bundle, R_MAX, threads=16)
    compare(got, ref, _tol(kind), "fuzz gpu %d %s" % (seed, kind))


# A cube splitter between two mirrors, one tilted by 0.003 deg: every pass through the cube sheds one beam sideways and the cavity beam
# walks off only after ~90 passes, so each root's beam tree is ~87 levels deep although it has only ~170 beams.  The node order
# (bundle order x breadth-first per tree) must hold for any depth (a random sweep found the engine's old 26 / 64-level key limits).
DEEP = -1  # marker for the cavity scene in mixed case lists


def cavity_case(kind, n):
    a = bmo.RoundPlanoMirror(30 * mm, 5 * mm)
    bmo.translate3d(a, [0, -30 * mm, 0])
    cube = bmo.CubeBeamsplitter(20 * mm, lambda lam: 1.5)
    b = bmo.RoundPlanoMirror(30 * mm, 5 * mm)
    bmo.zrotate3d(b, math.radians(180))
    bmo.xrotate3d(b, math.radians(0.003))
    bmo.translate3d(b, [0, 40 * mm, 0])
    system = bmo.System([a, cube, b])
    if kind == "ray":
        bundle = disc_bundle(n, [0, -20 * mm, 0], [0, 1, 0], 2 * mm, jitter=1e-4)
    else:
        bundle = gaussian_bundle(n, [0, -20 * mm, 0], [0, 1, 0], 2 * mm, w0=200e-6, jitter=1e-4)
    return bmo.CompiledScene(system, bundle.lambdas), bundle


def _tree_depth(res):
    depth = np.zeros(res.n_nodes, dtype=np.int64)
    for i in range(res.n_nodes):  # canonical order: parents come before children
        if res.node_parent[i] >= 0:
            depth[i] = depth[res.node_parent[i]] + 1
    return int(depth.max())


def test_deep_beam_tree_lane_code(oracle):
    scene, bundle = cavity_case("ray", 8)
    ref = oracle.trace(scene, bundle, R_MAX, threads=4)
    assert _tree_depth(ref) > 64
    compare(emu_trace(scene, bundle, R_MAX), ref, 0.0, "deep tree")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ray", "gauss"])
def test_deep_beam_tree_engine(oracle, kind):
    scene, bundle = cavity_case(kind, 300)
    ref = oracle.trace(scene, bundle, R_MAX, threads=16)
    assert _tree_depth(ref) > 64
    eng = bmo.Engine(scene, 0)
    try:
        got = eng.trace(bundle, R_MAX)
    finally:
        eng.close()
    compare(got, ref, _tol(kind), "deep tree gpu " + kind)


# ---------------------------------------------------------------------------------------------------------------- retrace
RETRACE_FUZZ = [(seed, "ray") for seed in range(401, 425)] + [(seed, "gauss") for seed in range(451, 459)] + [(seed, "pol") for seed in range(471, 479)]


def _retrace_case(seed, kind, n):
    """(scene before, scene after a small random move of one element, bundle)."""
    system, rng = random_system(seed, with_detectors=(kind != "gauss"))
    bundle = random_bundle(rng, kind, n)
    scene0 = bmo.CompiledScene(system, bundle.lambdas)
    objs = system.objects()
    target = objs[int(rng.integers(0, len(objs)))]
    if rng.random() < 0.5:
        bmo.translate3d(target, [float(rng.uniform(-0.5, 0.5) * mm), float(rng.uniform(-0.5, 0.5) * mm), float(rng.uniform(-0.5, 0.5) * mm)])
    else:
        bmo.xrotate3d(target, math.radians(float(rng.uniform(-1, 1))))
        bmo.zrotate3d(target, math.radians(float(rng.uniform(-1, 1))))
    scene1 = bmo.CompiledScene(system, bundle.lambdas)
    return scene0, scene1, bundle


@pytest.mark.parametrize("seed,kind", RETRACE_FUZZ)
def test_lane_code_retrace_equals_oracle_on_random_moves(oracle, seed, kind):
    scene0, scene1, bundle = _retrace_case(seed, kind, 96 if kind == "ray" else 48)
    e0 = _engine_first(lambda: emu_trace(scene0, bundle, R_MAX, max_beams=_limit(bundle.n)))
    e1 = _engine_first(lambda: emu_trace(scene1, bundle, R_MAX, prev=e0, max_beams=_limit(bundle.n)))
    a0, sol = oracle.trace(scene0, bundle, R_MAX, threads=4, keep=True)
    a1 = oracle.trace(scene1, bundle, R_MAX, threads=4, prev=sol)
    compare(e0, a0, _tol(kind), "retrace fuzz %d %s first" % (seed, kind))
    compare(e1, a1, _tol(kind), "retrace fuzz %d %s retrace" % (seed, kind))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kind", RETRACE_FUZZ)
def test_engine_retrace_equals_oracle_on_random_moves(oracle, seed, kind):
    scene0, scene1, bundle = _retrace_case(seed, kind, 2048 if kind == "ray" else 512)
    g0, h0 = _engine_first(lambda: bmo.system._engine_solve(scene0, bundle, R_MAX, None, max_beams=_limit(bundle.n)))
    g1, h1 = _engine_first(lambda: bmo.system._engine_solve(scene1, bundle, R_MAX, h0, max_beams=_limit(bundle.n)))
    a0, sol = oracle.trace(scene0, bundle, R_MAX, threads=16, keep=True)
    a1 = oracle.trace(scene1, bundle, R_MAX, threads=16, prev=sol)
    compare(g0, a0, _tol(kind), "retrace fuzz gpu %d %s first" % (seed, kind))
    compare(g1, a1, _tol(kind), "retrace fuzz gpu %d %s retrace" % (seed, kind))
    h0.free()
    h1.free()


# ---------------------------------------------------------------------------------------------------------------- Photodetector read-out
def _pd_seeds(lo, hi):
    """Gaussian draws whose train ends in a Photodetector (the others end in a mirror and have no detector)."""
    return [s for s in range(lo, hi) if random_system(s, with_detectors=False)[0].objects()[-1].__class__ is bmo.Photodetector]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", _pd_seeds(301, 341))
def test_engine_photodetector_field_on_random_scenes(oracle, seed):
    """Random Gaussian bundles through random trains onto a tilted Photodetector: recorded beamlets and the accumulated complex
    field (bmo_photodetector_field) against the oracle."""
    scene, bundle = _case(seed, "gauss", 512)
    if not scene.detectors:
        pytest.skip("train ends in a mirror: no detector in this draw")
    g, gsol = _engine_first(lambda: bmo.system._engine_solve(scene, bundle, R_MAX, None, max_beams=_limit(bundle.n)))
    a, osol = oracle.trace(scene, bundle, R_MAX, threads=16, keep=True)
    compare(g, a, 0.0, "pd fuzz %d" % seed)
    pd = scene.detectors[0]
    fa = np.zeros((len(pd.x), len(pd.y)), dtype=np.complex128)
    fg = fa.copy()
    osol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fa)
    gsol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fg)
    peak = np.abs(fa).max()
    if int(a.det_count[0]) > 0:
        assert peak > 0
    assert np.abs(fg - fa).max() <= 1e-9 * max(peak, 1e-300)
    gsol.free()
    osol.free()


# ---------------------------------------------------------------------------------------------------------------- kernel variants
VARIANTS = [{"BMO_NO_LDS": "1"}, {"BMO_FUSE": "1"}, {"BMO_FUSE": "3"}, {"BMO_FORCE_DEEP_ORDER": "1"}, {"BMO_FORCE_SORT_ORDER": "1"}, {"BMO_NO_LDS": "1", "BMO_FUSE": "2"}]


@pytest.mark.gpu
@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_engine_variants_on_random_scenes(oracle, monkeypatch, env):
    """The engine's other code paths — scene tables read from global memory instead of LDS (what a scene > 120 KB gets), other
    fused-level counts, the radix-sort and the level-by-level node ordering of trees deeper than 5 / 26 levels — on a slice of the random scenes, incl. a retrace."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for seed, kind in CASES[0:96:8] + [(DEEP, "ray")]:
        if seed == DEEP:
            scene, bundle = cavity_case("ray", 64)
        else:
            scene, bundle = _case(seed, kind, 1024 if kind == "ray" else 256)
        eng = bmo.Engine(scene, 0, max_beams=_limit(bundle.n))
        try:
            got = _engine_first(lambda: eng.trace(bundle, R_MAX))
        finally:
            eng.close()
        ref = oracle.trace(scene, bundle, R_MAX, threads=16)
        compare(got, ref, _tol(kind), "variant %s seed %s %s" % (env, seed, kind))
    scene0, scene1, bundle = _retrace_case(401, "ray", 1024)
    g0, h0 = bmo.system._engine_solve(scene0, bundle, R_MAX, None)
    g1, h1 = bmo.system._engine_solve(scene1, bundle, R_MAX, h0)
    a0, sol = oracle.trace(scene0, bundle, R_MAX, threads=16, keep=True)
    compare(g1, oracle.trace(scene1, bundle, R_MAX, threads=16, prev=sol), 0.0, "variant %s retrace" % env)
    h0.free()
    h1.free()


@pytest.mark.gpu
def test_scene_larger_than_lds(oracle):
    """A 2 400-triangle disc mirror makes the scene tables 190 KB: they stay in global memory (LDS holds 160 KB)."""
    mirror = bmo.Mirror(bmo.CircularFlatMesh(20 * mm, 2400))
    bmo.xrotate3d(mirror, math.radians(20))
    bmo.translate3d(mirror, [0, 60 * mm, 0])
    lens = bmo.SphericalLens(60 * mm, -60 * mm, 6 * mm, D, 1.5)
    det = bmo.Spotdetector(60 * mm)
    bmo.xrotate3d(det, math.radians(90))
    bmo.translate3d(det, [0, 0, -50 * mm])
    system = bmo.System([lens, mirror, det])
    bundle = disc_bundle(2048, [0, -20 * mm, 0], [0, 1, 0], 0.6 * D, jitter=0.01)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    assert scene.desc.n_tris > 2400
    eng = bmo.Engine(scene, 0)
    try:
        got = eng.trace(bundle, R_MAX)
    finally:
        eng.close()
    ref = oracle.trace(scene, bundle, R_MAX, threads=16)
    assert int(ref.det_count.sum()) > 0
    compare(got, ref, 0.0, "big scene")
