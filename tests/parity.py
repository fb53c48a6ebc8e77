"""Helpers shared by parity tests: run oracle / emulator / engine on one compiled scene and compare TraceResults."""
import ctypes as C
import os

import numpy as np

import bmo_amd as bmo
from bmo_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libbmo_emu.so")
_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        import subprocess

        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EMU)])
        _emu = C.CDLL(EMU)
        dp = C.POINTER(C.c_double)
        _emu.bmo_emu_jl_trig_n.restype = None
        _emu.bmo_emu_jl_trig_n.argtypes = [C.c_int, dp, dp, C.c_longlong, dp]
        _emu.bmo_emu_trace.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RayBatch), C.POINTER(abi.TraceOpts), C.POINTER(C.c_void_p),
                                       C.POINTER(abi.ResultView)]
        _emu.bmo_emu_free.argtypes = [C.c_void_p]
        _emu.bmo_emu_retrace.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RayBatch), C.POINTER(abi.TraceOpts), C.POINTER(abi.ResultView),
                                         C.POINTER(C.c_void_p), C.POINTER(abi.ResultView)]
    return _emu


def emu_jl_trig(which, x, y=None):
    """The lane code's restatement of Julia Base's sin / cos / tan / acos / atan / atan(y, x) (csrc/bmo_jlmath.hpp), host build."""
    L = emu_lib()
    dp = C.POINTER(C.c_double)
    xs = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
    ys = None if y is None else np.ascontiguousarray(np.atleast_1d(y), dtype=np.float64)
    out = np.zeros(xs.shape[0])
    L.bmo_emu_jl_trig_n(int(which), xs.ctypes.data_as(dp), None if ys is None else ys.ctypes.data_as(dp), xs.shape[0], out.ctypes.data_as(dp))
    return out


def emu_trace(scene, bundle, r_max=100, prev=None, max_beams=0):
    """Host build of the engine's lane code (tests/emu) — test-only.  prev = TraceResult of the previous solve => retrace."""
    _emu = emu_lib()
    batch, keep = bmo.make_batch(scene, bundle)
    o = abi.TraceOpts()
    o.r_max, o.device, o.record_segments, o.max_beams = int(r_max), 0, 1, int(max_beams)
    h = C.c_void_p()
    v = abi.ResultView()
    if prev is None:
        rc = _emu.bmo_emu_trace(C.byref(scene.desc), C.byref(batch), C.byref(o), C.byref(h), C.byref(v))
    else:
        pv = prev.as_view()
        rc = _emu.bmo_emu_retrace(C.byref(scene.desc), C.byref(batch), C.byref(o), C.byref(pv), C.byref(h), C.byref(v))
    if rc == -6:
        raise RuntimeError("emulator: beam tree exceeds max_beams (-6)")
    assert rc == 0, rc
    try:
        return abi.TraceResult(v)
    finally:
        _emu.bmo_emu_free(h)


def compare(a, b, rtol=0.0, label=""):
    """Assert two TraceResults agree: structure/ids/counts bit-exact, FP64 planes within rtol (0 = bit-exact)."""
    assert a.n_roots == b.n_roots, label
    assert a.n_nodes == b.n_nodes, (label, a.n_nodes, b.n_nodes)
    assert a.n_records == b.n_records, (label, a.n_records, b.n_records)
    for name in ("node_root", "node_parent", "node_first_child", "node_first_rec", "node_nseg", "node_status", "rec_obj", "rec_shape",
                 "det_count", "det_offset", "det_node"):
        x, y = getattr(a, name), getattr(b, name)
        assert np.array_equal(x, y), (label, name, np.flatnonzero(x != y)[:10])
    assert a.n_intersect_calls == b.n_intersect_calls, (label, a.n_intersect_calls, b.n_intersect_calls)
    for name in ("rec", "det_data", "node_aux"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape, (label, name)
        if rtol == 0.0:
            # bit patterns, not values: -0.0 == +0.0 compares equal although Base.min / max and isless tell them apart (NaN payloads aside)
            xb, yb = np.ascontiguousarray(x, dtype=np.float64).view(np.int64), np.ascontiguousarray(y, dtype=np.float64).view(np.int64)
            same = (xb == yb) | (np.isnan(x) & np.isnan(y))
            if not same.all():
                idx = np.argwhere(~same)[:5]
                raise AssertionError((label, name, "not bit-exact", idx.tolist(), x[~same][:5], y[~same][:5]))
        else:
            fin = np.isfinite(y)
            assert np.array_equal(fin, np.isfinite(x)), (label, name, "finite mask")
            err = np.abs(x[fin] - y[fin])
            tol = rtol * np.maximum(np.abs(y[fin]), 1e-3)  # relative, with a 1 mm floor for values that cross zero
            assert (err <= tol).all(), (label, name, float((err / np.maximum(np.abs(y[fin]), 1e-3)).max()))
    return True
