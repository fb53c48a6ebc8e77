"""GPU parity tests proper: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: hit object/shape ids, segment counts, node tree, status bits, detector order and the reference
intersect3d call count bit-exact; FP64 planes bit-exact for geometric rays (same operation order,
-ffp-contract=off on both sides) — for every beam kind since round 4: the step path's sin / cos / tan / acos / atan are Julia Base's own
algorithms, restated once for the oracle and once for the engine (tests/test_jl_trig.py), so no C library enters a trace.
"""
import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare
from scenes import c1_bundle, c1_scene, c2_bundle, c2_scene, c3_bundle, c4_bundle, c4_scene, c5_bundle, c5_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_ok():
    lib = bmo.abi.load_engine()
    assert lib.bmo_device_count() >= 1, "no HIP device visible"
    return lib


def run_both(oracle, system, bundle, r_max=100, threads=16):
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    try:
        got = eng.trace(bundle, r_max)
    finally:
        eng.close()
    ref = oracle.trace(scene, bundle, r_max, threads=threads)
    return got, ref


def test_device_selftest(engine_ok):
    """The hardware min / max the lane code uses (v_max_f64 / v_min_f64 + NaN select, and the Dual forms) against the rule written with
    compares (Base.max / Base.min: NaN-propagating, -0.0 < +0.0), bit for bit on the special values."""
    rc = engine_ok.bmo_selftest(0)
    assert rc == 0, engine_ok.bmo_last_error().decode()


def test_c1_bit_exact(engine_ok, oracle):
    system, _ = c1_scene()
    got, ref = run_both(oracle, system, c1_bundle(1000))
    compare(got, ref, 0.0, "c1")
    deepest = int(ref.node_nseg.max())
    assert (deepest + 31) // 32 <= got.n_steps <= deepest  # launches: up to 32 bounce levels are fused into one


def test_c2_bit_exact(engine_ok, oracle):
    system, _ = c2_scene()
    got, ref = run_both(oracle, system, c2_bundle(3000))
    compare(got, ref, 0.0, "c2")
    assert got.det_count.sum() > 0


@pytest.mark.parametrize("n", [1, 63, 65, 191, 257, 321, 4096 + 129])
def test_ragged_batch_sizes_read_the_segment_log(engine_ok, oracle, n):
    """Batch sizes that are not a multiple of the workgroup (m % 256 in 1..192): the tail waves of the grid have no record
    but still pass the fused level loop and note how far they got (Chunk::wl); the segment log of every beam must come out whole."""
    system, _ = c2_scene()
    got, ref = run_both(oracle, system, c2_bundle(n))
    compare(got, ref, 0.0, f"c2 n={n}")
    assert got.n_records == int(ref.node_nseg.sum())


@pytest.mark.parametrize("r_max", [-3, 0, 1, 2, 3, 10])
def test_r_max_cap(engine_ok, oracle, r_max):
    system, _ = c2_scene()
    got, ref = run_both(oracle, system, c2_bundle(256), r_max=r_max)
    compare(got, ref, 0.0, f"rmax{r_max}")
    assert int(got.node_nseg.max()) <= max(r_max, 1)


def test_empty_batch(engine_ok, oracle):
    system, _ = c1_scene()
    b = c1_bundle(4)
    b = bmo.RayBundle(b.kind, b.planes[:, :0])
    scene = bmo.CompiledScene(system, [1.064e-6])
    eng = bmo.Engine(scene, 0)
    got = eng.trace(b)
    eng.close()
    assert got.n_nodes == 0 and got.n_records == 0 and got.n_steps == 0


# Rounds 1 - 3 compared polarized / Gaussian traces at north_star's 1e-10: acos / sin / cos (Fresnel, P-matrix) and tan / acos / atan
# (gauss_parameters) came from ocml on the device and glibc in the oracle.  Round 4: both sides evaluate Julia Base's own algorithms
# (csrc/bmo_jlmath.hpp, oracle/jl_trig.hpp), so the tolerance of a trace is zero for every beam kind.  (The read-outs — PSF, Photodetector
# field — still call sincos / exp of the platform's library on phases of 10^5 rad and keep 1e-10.)
LIBM_RTOL = 0.0


def test_c4_polarized(engine_ok, oracle):
    system, _ = c4_scene()
    got, ref = run_both(oracle, system, c4_bundle(4096))
    compare(got, ref, LIBM_RTOL, "c4")
    assert got.rec_planes == 17 and int(got.node_nseg.max()) == 7


def test_c3_gaussian(engine_ok, oracle):
    system, _ = c2_scene()
    got, ref = run_both(oracle, system, c3_bundle(1024))
    compare(got, ref, LIBM_RTOL, "c3")
    assert got.rec_planes == 33 and got.n_nodes == 3 * got.n_roots
    # geometric planes of all three rays are libm-free and must be bit-exact
    assert np.array_equal(got.rec, ref.rec)


def test_c5_32_elements(engine_ok, oracle):
    system, _ = c5_scene()
    got, ref = run_both(oracle, system, c5_bundle(1536))
    compare(got, ref, 0.0, "c5")
    assert got.det_count.sum() > 0


def test_aspheres_and_cylinder_lenses(engine_ok, oracle):
    from scenes import disc_bundle
    from test_oracle_kat3 import asphere_cylinder_scene

    b = disc_bundle(2048, center=[0, -0.05, 0], direction=[0, 1, 0], diameter=0.044, e1=[1, 0, 0], jitter=5e-3)
    got, ref = run_both(oracle, asphere_cylinder_scene(), b, r_max=40)
    compare(got, ref, 0.0, "asph+cyl")


def test_acylinder_lenses(engine_ok, oracle):
    import math

    from scenes import disc_bundle, mm
    from test_oracle_kat3 import AYL

    a1 = bmo.Lens(bmo.AcylindricalSurface(15.538e-3, 25e-3, 50e-3, -1.0, AYL), 7.5e-3, lambda n: 1.777)
    a2 = bmo.Lens(bmo.AcylindricalSurface(-15.538e-3, 25e-3, 50e-3, -1.0, AYL), 7.5e-3, lambda n: 1.6)
    bmo.translate3d(a2, [0, 20 * mm, 0])
    bmo.yrotate3d(a2, math.radians(35))
    det = bmo.Spotdetector(80 * mm)
    bmo.translate3d(det, [0, 60 * mm, 0])
    b = disc_bundle(2048, center=[0, -0.03, 0], direction=[0, 1, 0], diameter=22 * mm, e1=[1, 0, 0], jitter=5e-3)
    got, ref = run_both(oracle, bmo.System([a1, a2, det]), b, r_max=40)
    compare(got, ref, 0.0, "acyl")


def test_split_form_and_host_hit_copy(engine_ok, oracle):
    """bmo_batch_upload + bmo_trace_device + bmo_result_copy_hits into HOST memory + bmo_result_view give the same tables as
    the one-call bmo_trace; a second view of a same-sized result (pinned tables reused from the pool) is identical too."""
    system, _ = c2_scene()
    bundle = c2_bundle(4096)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    ref = oracle.trace(scene, bundle, 100, threads=8)
    eng = bmo.Engine(scene, 0)
    try:
        dev = eng.upload(bundle)
        for rep in range(2):
            res = eng.trace_device(dev, 100)
            for slot in range(len(scene.detectors)):
                _, cnt = eng.result_device_hits(res, slot)
                assert cnt == int(ref.det_count[slot])
                host = np.full((cnt + 1, 9), -7.0)
                eng.result_copy_hits(res, slot, host.ctypes.data, cnt)
                assert np.array_equal(host[:cnt], ref.detector_hits(slot)) and np.all(host[cnt] == -7.0)
            got = eng.result_view(res)
            eng.free_result(res)
            compare(got, ref, 0.0, "split form rep %d" % rep)
        eng.free_batch(dev)
    finally:
        eng.close()


def test_max_beams_limit(engine_ok):
    """bmo_trace_opts.max_beams stops a solve whose beam tree outgrows the limit with BMO_ERR_LIMIT; 0 / a generous limit do not."""
    system, _ = c2_scene()
    bundle = c2_bundle(1024)  # one splitter: 3 beams per ray
    scene = bmo.CompiledScene(system, bundle.lambdas)
    for limit, ok in ((0, True), (3 * 1024, True), (2000, False)):
        eng = bmo.Engine(scene, 0, max_beams=limit)
        try:
            if ok:
                assert eng.trace(bundle, 100).n_nodes == 3 * 1024
            else:
                with pytest.raises(RuntimeError, match=r"\(-6\).*max_beams"):
                    eng.trace(bundle, 100)
        finally:
            eng.close()


def test_concurrent_host_threads(engine_ok, oracle):
    """include/bmo.h "Threading": trace calls may come from several host threads (they take turns on a device), one scene may be
    shared; every thread gets its own, correct solution."""
    import threading

    system, _ = c2_scene()
    bundles = [c2_bundle(2048 + 512 * i) for i in range(4)]
    scene = bmo.CompiledScene(system, bundles[0].lambdas)
    refs = [oracle.trace(scene, b, 100, threads=8) for b in bundles]
    eng = bmo.Engine(scene, 0)  # one scene handle shared by all threads
    out, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                out[i] = eng.trace(bundles[i], 100)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    eng.close()
    assert not errs, errs
    for i in range(4):
        compare(out[i], refs[i], 0.0, "thread %d" % i)


def test_argument_validation(engine_ok):
    """Bad descriptors come back as BMO_ERR_INVALID with a message, never as a fault on the device."""
    import ctypes as C

    from bmo_amd import abi
    from bmo_amd.system import make_batch

    lib = engine_ok
    system, _ = c1_scene()
    bundle = c1_bundle(16)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    try:
        o = eng.opts(100)
        res = C.c_void_p()

        def call(batch, opts=o):
            return lib.bmo_trace(eng.handle, C.byref(batch), C.byref(opts), C.byref(res))

        batch, keep = make_batch(scene, bundle)
        assert call(batch) == 0
        lib.bmo_result_free(res)
        batch.kind = 7
        assert call(batch) == -1 and b"kind" in lib.bmo_last_error()
        batch, keep = make_batch(scene, bundle)
        batch.n_planes -= 1
        assert call(batch) == -1 and b"plane" in lib.bmo_last_error()
        batch, keep = make_batch(scene, bundle)
        batch.n = -5
        assert call(batch) == -1
        batch, keep = make_batch(scene, bundle)
        idx = np.full(bundle.n, 3, dtype=np.int32)  # only one wavelength in this scene
        batch.lambda_idx = idx.ctypes.data_as(C.POINTER(C.c_int32))
        assert call(batch) == -1 and b"lambda" in lib.bmo_last_error()
        batch, keep = make_batch(scene, bundle)
        bad = eng.opts(100)
        bad.device = 99
        assert call(batch, bad) == -1 and b"device" in lib.bmo_last_error()
        assert lib.bmo_trace(None, C.byref(batch), C.byref(o), C.byref(res)) == -1
        # retrace against a batch of another size / kind
        assert call(batch) == 0
        prev = C.c_void_p(res.value)
        other, keep2 = make_batch(scene, c1_bundle(8))
        out = C.c_void_p()
        assert lib.bmo_retrace(eng.handle, C.byref(other), prev, C.byref(o), C.byref(out)) == -1 and b"retrace" in lib.bmo_last_error()
        lib.bmo_result_free(prev)
    finally:
        eng.close()


@pytest.mark.parametrize("kind", ["ray", "gauss"])
def test_record_segments_off(engine_ok, oracle, kind):
    """bmo_trace_opts.record_segments = 0: beam tree, statuses, counts and detector hits as with the log; the log itself is not
    kept (view reports no records), and what needs it (retrace, Photodetector field) refuses."""
    import ctypes as C

    system, _ = c2_scene()
    bundle = c2_bundle(3000) if kind == "ray" else c3_bundle(600)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    ref = oracle.trace(scene, bundle, 100, threads=8)
    eng = bmo.Engine(scene, 0)
    try:
        dev = eng.upload(bundle)
        full = eng.trace_device(dev, 100)
        lean = eng.trace_device(dev, 100, record_segments=False)
        assert eng.result_size(lean) == eng.result_size(full) == (ref.n_intersect_calls, ref.n_records, ref.n_nodes, int(ref.det_count.sum()))
        vf, vl = eng.result_view(full), eng.result_view(lean)
        for name in ("node_root", "node_parent", "node_first_child", "node_nseg", "node_status", "det_count", "det_offset", "det_node"):
            assert np.array_equal(getattr(vf, name), getattr(vl, name)), name
        assert np.array_equal(vf.det_data, vl.det_data) and np.array_equal(vf.node_aux, vl.node_aux, equal_nan=True)
        assert vl.n_records == 0 and vl.rec.size == 0 and vf.n_records == ref.n_records
        out = C.c_void_p()
        o = eng.opts(100)
        assert eng.lib.bmo_retrace_device(eng.handle, dev, lean, C.byref(o), C.byref(out)) == -1 and b"record_segments" in eng.lib.bmo_last_error()
        eng.free_result(full)
        eng.free_result(lean)
        eng.free_batch(dev)
    finally:
        eng.close()


@pytest.mark.parametrize("kind", ["ray", "gauss"])
def test_full_size_properties(engine_ok, oracle, kind):
    """BASELINE's full size (config C2, 2^20 rays; 19.9 M segments) through size-independent properties: determinism, the sharding
    identity of SURVEY 8e (hit tables of contiguous shards concatenate to the un-sharded tables, bit for bit), the counters'
    internal consistency, and a strided 1024-ray sample of the full solve against the oracle."""
    n = 1 << 20
    system, _ = c2_scene()
    bundle = c2_bundle(n) if kind == "ray" else c3_bundle(n)  # C2 / C3
    rows = 1 if kind == "ray" else 3  # detector records per beam (Gaussian: chief, waist, divergence)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    try:
        def solve(b, keep_log):
            dev = eng.upload(b)
            res = eng.trace_device(dev, 100, record_segments=keep_log)
            size = eng.result_size(res)
            v = eng.result_view(res)  # without the log this is the beam tree + hit tables only
            eng.free_result(res)
            eng.free_batch(dev)
            return size, v

        size_full, full = solve(bundle, False)
        size_again, again = solve(bundle, kind == "ray")  # the logging solve (Ray): same beams and hits, its record count is checked below
        assert size_full == size_again
        calls, nrec, nnodes, nhits = size_full
        assert nnodes == full.n_nodes == 3 * n and nhits == int(full.det_count.sum()) == 2 * n * rows  # one splitter, both arms detected
        assert int(full.node_nseg.astype(np.int64).sum()) == nrec
        if kind == "ray":
            assert again.n_records == nrec
        for name in ("node_root", "node_parent", "node_nseg", "node_status", "det_count", "det_node"):
            assert np.array_equal(getattr(full, name), getattr(again, name)), name
        assert np.array_equal(full.det_data, again.det_data)
        # sharding identity
        half = n // 2
        parts = [solve(bmo.RayBundle(bundle.kind, bundle.planes[:, lo:lo + half]), False) for lo in (0, half)]
        assert sum(p[0][0] for p in parts) == calls and sum(p[0][1] for p in parts) == nrec
        for slot in range(len(scene.detectors)):
            assert np.array_equal(np.concatenate([p[1].detector_hits(slot) for p in parts]), full.detector_hits(slot)), slot
        # a strided sample against the oracle
        idx = np.arange(0, n, 1024)
        sample = bmo.RayBundle(bundle.kind, bundle.planes[:, idx])
        ref = oracle.trace(scene, sample, 100, threads=16)
        pick = np.isin(full.node_root, idx)
        assert np.array_equal(full.node_nseg[pick], ref.node_nseg) and np.array_equal(full.node_status[pick], ref.node_status)
        for slot in range(len(scene.detectors)):
            lo, cnt = int(full.det_offset[slot]), int(full.det_count[slot])
            roots_of_hits = full.node_root[full.det_node[lo:lo + cnt]]
            mine, theirs = full.detector_hits(slot)[np.isin(roots_of_hits, idx)], ref.detector_hits(slot)
            if kind == "ray":
                assert np.array_equal(mine, theirs), slot
            else:
                assert mine.shape == theirs.shape and np.allclose(mine, theirs, rtol=1e-10, atol=0), slot
    finally:
        eng.close()


def test_selective_views(engine_ok, oracle):
    """bmo_result_view_select: node tables always; hits, the last segment of every beam (last(rays(beam)), Beam.jl:79) or the whole
    log on request — each equal to the matching part of the full view, which equals the oracle."""
    from bmo_amd import abi

    system, _ = c2_scene()
    for bundle in (c2_bundle(3000), c3_bundle(500)):
        scene = bmo.CompiledScene(system, bundle.lambdas)
        ref = oracle.trace(scene, bundle, 100, threads=8)
        eng = bmo.Engine(scene, 0)
        try:
            dev = eng.upload(bundle)
            res = eng.trace_device(dev, 100)
            nodes = eng.result_view(res, 0)
            assert nodes.n_records == 0 and nodes.rec.size == 0 and nodes.det_data.size == 0
            assert np.array_equal(nodes.det_count, ref.det_count)
            last = eng.result_view(res, abi.VIEW_HITS | abi.VIEW_LAST_SEGMENT)
            assert last.n_records == last.n_nodes == ref.n_nodes
            assert np.array_equal(last.node_first_rec, np.arange(ref.n_nodes))
            full = eng.result_view(res, abi.VIEW_HITS | abi.VIEW_SEGMENTS)
            compare(full, ref, 0.0 if bundle.kind == 0 else LIBM_RTOL, "full view after narrower ones")
            for v in (nodes, last):
                for name in ("node_root", "node_parent", "node_first_child", "node_nseg", "node_status", "det_count", "det_offset"):
                    assert np.array_equal(getattr(v, name), getattr(full, name)), name
                assert np.array_equal(v.node_aux, full.node_aux, equal_nan=True)
            at = full.node_first_rec + full.node_nseg - 1
            assert np.array_equal(last.rec, full.rec[:, at], equal_nan=True)
            assert np.array_equal(last.rec_obj, full.rec_obj[at]) and np.array_equal(last.rec_shape, full.rec_shape[at])
            assert np.array_equal(last.det_data, full.det_data) and np.array_equal(last.det_node, full.det_node)
            import ctypes as C

            v = abi.ResultView()  # the whole log is on the host now: a LAST-only request is refused, not silently re-shaped
            assert eng.lib.bmo_result_view_select(res, abi.VIEW_LAST_SEGMENT, C.byref(v)) == -1
            eng.free_result(res)
            eng.free_batch(dev)
        finally:
            eng.close()


def test_hit_columns_and_empty_copies(engine_ok, oracle):
    """bmo_result_copy_hit_columns packs the leading columns into host or device memory; a detector without hits accepts a NULL
    destination (torch.empty((0, 9)).data_ptr() == 0) in both copy calls."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    system, _ = c2_scene()
    bundle = c2_bundle(2048)
    bundle.planes[3:6, :] = np.array([[1.0], [0.0], [0.0]])  # every ray leaves sideways: no detector is reached
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    try:
        res = eng.trace_device(eng.upload(bundle), 100)
        for slot in range(len(scene.detectors)):
            assert eng.result_device_hits(res, slot)[1] == 0
            eng.result_copy_hits(res, slot, 0, 0)
            eng.result_copy_hit_columns(res, slot, 2, 0, 0)
        eng.free_result(res)
    finally:
        eng.close()
    bundle = c2_bundle(2048)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    ref = oracle.trace(scene, bundle, 100, threads=8)
    eng = bmo.Engine(scene, 0)
    try:
        res = eng.trace_device(eng.upload(bundle), 100)
        for slot in range(len(scene.detectors)):
            cnt = eng.result_device_hits(res, slot)[1]
            want = ref.detector_hits(slot)
            assert cnt == len(want) > 0
            for cols in (2, 5, 9):
                host = np.full((cnt + 1, cols), -3.0)
                eng.result_copy_hit_columns(res, slot, cols, host.ctypes.data, cnt)
                assert np.array_equal(host[:cnt], want[:, :cols]) and np.all(host[cnt] == -3.0)
                dptr = C.c_void_p()
                assert hip.hipMalloc(C.byref(dptr), C.c_size_t(cnt * cols * 8)) == 0
                eng.result_copy_hit_columns(res, slot, cols, dptr.value, cnt)
                back = np.zeros((cnt, cols))
                assert hip.hipMemcpy(C.c_void_p(back.ctypes.data), dptr, C.c_size_t(cnt * cols * 8), 2) == 0  # hipMemcpyDeviceToHost
                hip.hipFree(dptr)
                assert np.array_equal(back, want[:, :cols])
        with pytest.raises(RuntimeError):
            eng.result_copy_hit_columns(res, 0, 10, 0, 1)
        eng.free_result(res)
    finally:
        eng.close()


@pytest.mark.parametrize("config", ["c4", "c5", "c2v", "c2s"])
def test_full_size_properties_c4_c5(engine_ok, oracle, config):
    """BASELINE configs 4 (2^18 PolarizedRays, 6 refracting surfaces) and 5 (one GPU's shard: 2^21 Rays, 32 elements) at FULL size
     through size-independent properties: determinism with and without the segment log, consistent counters, the
    sharding identity of SURVEY 8e on the hit tables of two contiguous halves, and a strided sample against the oracle (statuses and
    segment counts bit-exact; records bit-exact for config 5, 1e-10 relative for the polarized config 4)."""
    # (round 3) also the two config-2 bundles the bench reports besides the headline one, at 2^20 rays: "c2v", the ragged (vignetted)
    # bundle, and "c2s", SURVEY 8(d)'s literal bundle (Fibonacci disc 0.8 x the first aperture, along the axis, 2 mrad jitter).
    if config == "c4":
        n, (system, _), rtol = 1 << 18, c4_scene(), LIBM_RTOL
        bundle = c4_bundle(n)
    elif config == "c5":
        n, (system, _), rtol = 1 << 21, c5_scene(), 0.0
        bundle = c5_bundle(n)
    else:
        from scenes import c2_survey_bundle, c2_vignetted_bundle

        n, (system, _), rtol = 1 << 20, c2_scene(), 0.0
        bundle = c2_vignetted_bundle(n) if config == "c2v" else c2_survey_bundle(n)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    try:
        def solve(b, keep_log, view=True):
            dev = eng.upload(b)
            res = eng.trace_device(dev, 100, record_segments=keep_log)
            size = eng.result_size(res)
            v = eng.result_view(res, 1) if view else None  # node tables + hits, never the multi-GB log
            eng.free_result(res)
            eng.free_batch(dev)
            return size, v

        size_a, a = solve(bundle, True)
        size_b, b = solve(bundle, False)
        assert size_a == size_b
        calls, nrec, nnodes, nhits = size_a
        assert int(a.node_nseg.astype(np.int64).sum()) == nrec and nnodes == a.n_nodes and nhits == int(a.det_count.sum())
        for name in ("node_root", "node_parent", "node_first_child", "node_nseg", "node_status", "det_count", "det_node"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        assert np.array_equal(a.det_data, b.det_data)
        if config == "c4":
            assert nnodes == n and int(a.node_nseg.max()) == 7 and nhits == 0  # no splitter, no detector: 7 segments, then the end stop
        elif config == "c5":
            assert nnodes > n and nhits > n  # the middle train splits at the beamsplitter; most rays reach a detector
        else:
            assert nnodes > n and 0 < nhits <= 2 * n  # some roots reach the splitter, both arms end on detectors
            if config == "c2v":
                assert int((a.node_status & 1).sum()) > n // 100  # a ragged bundle: plenty of beams end in a miss
        half = n // 2
        parts = [solve(bmo.RayBundle(bundle.kind, bundle.planes[:, lo:lo + half]), False) for lo in (0, half)]
        assert sum(p[0][0] for p in parts) == calls and sum(p[0][1] for p in parts) == nrec and sum(p[0][2] for p in parts) == nnodes
        for slot in range(len(scene.detectors)):
            assert np.array_equal(np.concatenate([p[1].detector_hits(slot) for p in parts]), a.detector_hits(slot)), slot
        # strided sample against the oracle: its own small solve with the full records
        step = n // 512
        idx = np.arange(0, n, step)
        sample = bmo.RayBundle(bundle.kind, bundle.planes[:, idx])
        ref = oracle.trace(scene, sample, 100, threads=16)
        got = eng.trace(sample, 100)
        compare(got, ref, rtol, config + " sample")
        pick = np.isin(a.node_root, idx)
        assert np.array_equal(a.node_nseg[pick], ref.node_nseg) and np.array_equal(a.node_status[pick], ref.node_status)
        for slot in range(len(scene.detectors)):
            lo, cnt = int(a.det_offset[slot]), int(a.det_count[slot])
            roots_of_hits = a.node_root[a.det_node[lo:lo + cnt]]
            assert np.array_equal(a.detector_hits(slot)[np.isin(roots_of_hits, idx)], ref.detector_hits(slot)), slot
    finally:
        eng.close()
