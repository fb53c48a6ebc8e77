"""ctypes mirror of include/bmo.h and the loader of the HIP engine (libbmo_hip.so).

There is NO CPU fallback: `load_engine()` raises if the compiled HIP library is missing, and
every trace entry point of the library itself fails without a GPU (BMO_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 1
NPARAM = 8

VIEW_HITS, VIEW_LAST_SEGMENT, VIEW_SEGMENTS = 1, 2, 4  # bmo_result_view_select masks

PLANES_IN = {0: 8, 1: 14, 2: 25}
PLANES_REC = {0: 11, 1: 17, 2: 33}

NODE_MISS, NODE_STOPPED, NODE_RMAX, NODE_SPLIT, NODE_DETECTED, NODE_ERR_UNIT, NODE_GAUSS_DIVERGED, NODE_BLOCKED, NODE_ERR_ORTHO = (
    1, 2, 4, 8, 16, 32, 64, 128, 256)


class Shape(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("child_begin", C.c_int32), ("child_count", C.c_int32), ("tri_begin", C.c_int32),
        ("tri_count", C.c_int32), ("flags", C.c_int32),
        ("pos", C.c_double * 3), ("dir", C.c_double * 9), ("tdir", C.c_double * 9), ("p", C.c_double * NPARAM),
        ("bs_center", C.c_double * 3), ("bs_radius", C.c_double),
    ]


class Object(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("shape", C.c_int32 * 3), ("medium", C.c_int32 * 2), ("detector", C.c_int32), ("reserved", C.c_int32),
        ("reflectance", C.c_double), ("transmittance", C.c_double), ("cutoff", C.c_double), ("jones", C.c_double * 18),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_objects", C.c_int32), ("n_shapes", C.c_int32), ("n_children", C.c_int32),
        ("n_tris", C.c_int32), ("n_media", C.c_int32), ("n_lambda", C.c_int32), ("n_detectors", C.c_int32),
        ("objects", C.POINTER(Object)), ("shapes", C.POINTER(Shape)), ("children", C.POINTER(C.c_int32)),
        ("tris", C.POINTER(C.c_double)), ("n_table", C.POINTER(C.c_double)), ("lambdas", C.POINTER(C.c_double)),
        ("coefs", C.POINTER(C.c_double)),
        ("eps_srf", C.c_double), ("eps_ray", C.c_double), ("eps_ins", C.c_double), ("mt_keps", C.c_double),
        ("mt_leps", C.c_double), ("grad_h", C.c_double), ("march_iters", C.c_int32), ("n_coefs", C.c_int32),
    ]


class RayBatch(C.Structure):
    _fields_ = [("n", C.c_int64), ("kind", C.c_int32), ("n_planes", C.c_int32), ("planes", C.POINTER(C.c_double)),
                ("lambda_idx", C.POINTER(C.c_int32))]


class TraceOpts(C.Structure):
    _fields_ = [("r_max", C.c_int32), ("device", C.c_int32), ("record_segments", C.c_int32), ("max_beams", C.c_int32)]


class ResultView(C.Structure):
    _fields_ = [
        ("n_roots", C.c_int64), ("n_nodes", C.c_int64), ("n_records", C.c_int64), ("n_intersect_calls", C.c_int64),
        ("n_steps", C.c_int32), ("beam_kind", C.c_int32), ("rec_planes", C.c_int32), ("n_detectors", C.c_int32),
        ("node_root", C.POINTER(C.c_int32)), ("node_parent", C.POINTER(C.c_int32)), ("node_first_child", C.POINTER(C.c_int32)),
        ("node_first_rec", C.POINTER(C.c_int32)), ("node_nseg", C.POINTER(C.c_int32)), ("node_status", C.POINTER(C.c_int32)),
        ("node_aux", C.POINTER(C.c_double)),
        ("rec_obj", C.POINTER(C.c_int32)), ("rec_shape", C.POINTER(C.c_int32)), ("rec", C.POINTER(C.c_double)),
        ("det_count", C.POINTER(C.c_int64)), ("det_offset", C.POINTER(C.c_int64)), ("det_node", C.POINTER(C.c_int32)),
        ("det_data", C.POINTER(C.c_double)),
    ]


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class TraceResult:
    """Host copy of a bmo_trace_result_view (arrays are numpy copies; see include/bmo.h for the layout)."""

    def __init__(self, v):
        nn, nr = int(v.n_nodes), int(v.n_records)
        self.n_roots = int(v.n_roots)
        self.n_nodes = nn
        self.n_records = nr
        self.n_intersect_calls = int(v.n_intersect_calls)
        self.n_steps = int(v.n_steps)
        self.beam_kind = int(v.beam_kind)
        self.rec_planes = int(v.rec_planes)
        self.n_detectors = int(v.n_detectors)
        self.node_root = _np(v.node_root, nn, np.int32)
        self.node_parent = _np(v.node_parent, nn, np.int32)
        self.node_first_child = _np(v.node_first_child, nn, np.int32)
        self.node_first_rec = _np(v.node_first_rec, nn, np.int32)
        self.node_nseg = _np(v.node_nseg, nn, np.int32)
        self.node_status = _np(v.node_status, nn, np.int32)
        self.node_aux = _np(v.node_aux, nn * 4, np.float64).reshape(nn, 4)
        self.rec_obj = _np(v.rec_obj, nr, np.int32)
        self.rec_shape = _np(v.rec_shape, nr, np.int32)
        self.rec = _np(v.rec, nr * self.rec_planes, np.float64).reshape(self.rec_planes, nr)
        nd = self.n_detectors
        self.det_count = _np(v.det_count, nd, np.int64)
        self.det_offset = _np(v.det_offset, nd, np.int64)
        tot = int(self.det_count.sum())
        if not v.det_data:  # a selective view without BMO_VIEW_HITS: counts only
            tot = 0
        self.det_node = _np(v.det_node, tot, np.int32)
        self.det_data = _np(v.det_data, tot * 9, np.float64).reshape(tot, 9)

    def as_view(self):
        """A ResultView over this object's numpy arrays (valid while `self` is alive) — e.g. the `prev` of a test emulator."""
        v = ResultView()
        v.n_roots, v.n_nodes, v.n_records, v.n_intersect_calls = self.n_roots, self.n_nodes, self.n_records, self.n_intersect_calls
        v.n_steps, v.beam_kind, v.rec_planes, v.n_detectors = self.n_steps, self.beam_kind, self.rec_planes, self.n_detectors
        keep = []
        for name, ct in (("node_root", C.c_int32), ("node_parent", C.c_int32), ("node_first_child", C.c_int32), ("node_first_rec", C.c_int32),
                         ("node_nseg", C.c_int32), ("node_status", C.c_int32), ("node_aux", C.c_double), ("rec_obj", C.c_int32),
                         ("rec_shape", C.c_int32), ("rec", C.c_double), ("det_count", C.c_int64), ("det_offset", C.c_int64),
                         ("det_node", C.c_int32), ("det_data", C.c_double)):
            a = np.ascontiguousarray(getattr(self, name))
            keep.append(a)
            setattr(v, name, a.ctypes.data_as(C.POINTER(ct)))
        self._view_keep = keep
        return v

    def detector_hits(self, slot):
        o, c = int(self.det_offset[slot]), int(self.det_count[slot])
        return self.det_data[o:o + c]

    def detector_nodes(self, slot):
        o, c = int(self.det_offset[slot]), int(self.det_count[slot])
        return self.det_node[o:o + c]


_HERE = os.path.dirname(os.path.abspath(__file__))
ENGINE_PATH = os.environ.get("BMO_ENGINE_LIB") or os.path.join(_HERE, "csrc", "libbmo_hip.so")  # env: A/B builds of the HIP engine
_engine = None


class EngineMissing(RuntimeError):
    pass


def _check_source_hash(lib):
    """Refuse a library that was not built from the sources next to it (a stale, git-ignored .so pushed along with newer sources).
    Skipped for an explicitly chosen build (BMO_ENGINE_LIB: A/B runs) and where the sources are not there to compare with."""
    if os.environ.get("BMO_ENGINE_LIB"):
        return
    root = os.path.dirname(_HERE)
    srcs = [os.path.join(_HERE, "csrc", "bmo_engine.hip"), os.path.join(_HERE, "csrc", "bmo_lane.hpp"), os.path.join(_HERE, "csrc", "bmo_jlmath.hpp"),
            os.path.join(_HERE, "csrc", "bmo_readout.inc.hpp"), os.path.join(root, "include", "bmo.h")]
    if not all(os.path.exists(p) for p in srcs):
        return
    import hashlib

    h = hashlib.sha256()
    for p in srcs:
        h.update(open(p, "rb").read())
    try:
        lib.bmo_source_hash.restype = C.c_char_p
        built = lib.bmo_source_hash().decode()
    except AttributeError:
        built = None
    if built != h.hexdigest():
        raise EngineMissing(f"{ENGINE_PATH} was not built from the sources in this tree (source hash {built!r} != {h.hexdigest()[:16]}...): "
                            "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
    # ... nor one built with other compiler flags (bit parity needs -ffp-contract=off & co.): the build script's flag list decides
    want = _wanted_flags_hash(root)
    if want is not None:
        try:
            lib.bmo_build_flags_hash.restype = C.c_char_p
            got = lib.bmo_build_flags_hash().decode()
        except AttributeError:
            got = None
        if got != want:
            raise EngineMissing(f"{ENGINE_PATH} was built with other compiler flags than __graft_entry__.HIP_FLAGS (flags hash {got!r} != {want!r}): "
                                "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")


def _wanted_flags_hash(root):
    """flags_hash() of the build script next to the package (None where there is none: an installed copy of the package)."""
    path = os.path.join(root, "__graft_entry__.py")
    if not os.path.exists(path):
        return None
    import importlib.util

    spec = importlib.util.spec_from_file_location("_bmo_graft_entry", path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
        return mod.flags_hash()
    except Exception:
        return None


def load_engine():
    """Load the HIP engine.  Fails loudly when the compiled library is missing or stale."""
    global _engine
    if _engine is not None:
        return _engine
    if not os.path.exists(ENGINE_PATH):
        raise EngineMissing(
            f"{ENGINE_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(ENGINE_PATH)
    _check_source_hash(lib)
    vp = C.c_void_p
    lib.bmo_version.restype = C.c_int
    lib.bmo_last_error.restype = C.c_char_p
    lib.bmo_device_count.restype = C.c_int
    lib.bmo_selftest.argtypes = [C.c_int32]
    lib.bmo_selftest.restype = C.c_int
    lib.bmo_jl_trig.argtypes = [C.c_int32, C.c_double, C.c_double]
    lib.bmo_jl_trig.restype = C.c_double
    lib.bmo_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    lib.bmo_scene_destroy.argtypes = [vp]
    lib.bmo_trace.argtypes = [vp, C.POINTER(RayBatch), C.POINTER(TraceOpts), C.POINTER(vp)]
    lib.bmo_batch_upload.argtypes = [vp, C.POINTER(RayBatch), C.c_int32, C.POINTER(vp)]
    lib.bmo_batch_free.argtypes = [vp]
    lib.bmo_trace_device.argtypes = [vp, vp, C.POINTER(TraceOpts), C.POINTER(vp)]
    lib.bmo_result_device_hits.argtypes = [vp, C.c_int32, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int64)]
    lib.bmo_result_copy_hits.argtypes = [vp, C.c_int32, C.c_void_p, C.c_int64]
    lib.bmo_result_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.bmo_result_counts.argtypes = [vp] + [C.POINTER(C.c_int64)] * 4
    lib.bmo_result_view.argtypes = [vp, C.POINTER(ResultView)]
    lib.bmo_result_view_select.argtypes = [vp, C.c_uint32, C.POINTER(ResultView)]
    lib.bmo_result_copy_hit_columns.argtypes = [vp, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
    lib.bmo_result_free.argtypes = [vp]
    lib.bmo_retrace.argtypes = [vp, C.POINTER(RayBatch), vp, C.POINTER(TraceOpts), C.POINTER(vp)]
    lib.bmo_retrace_device.argtypes = [vp, vp, vp, C.POINTER(TraceOpts), C.POINTER(vp)]
    dp = C.POINTER(C.c_double)
    lib.bmo_photodetector_field.argtypes = [vp, C.c_int32, dp, dp, dp, dp, C.c_int32, C.c_int32, dp, dp]
    lib.bmo_gauss_parameters.argtypes = [vp, C.c_int64, dp, C.c_int32, dp]
    lib.bmo_result_set_gauss_prefix.argtypes = [vp, C.c_int64, C.POINTER(C.c_int32), dp, dp]
    lib.bmo_psf_intensity.argtypes = [C.c_void_p, C.c_int64, C.c_int32, dp, dp, dp, dp, dp, C.c_int32, C.c_int32, dp, dp, dp]
    _engine = lib
    return lib


def check(lib, rc, what):
    if rc != 0:
        msg = lib.bmo_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def psf_intensity(hits, origin, e1, e2, xs, zs, device=0, hits_device_ptr=None, n_hits=None, want_field=False):
    """bmo_psf_intensity: returns (I[n, n] indexed [i, j], field or None, kernel_ms).  `hits` is a host [H, 9] array, or pass
    `hits_device_ptr` + `n_hits` for a buffer already resident on `device` (bmo_result_device_hits)."""
    lib = load_engine()
    dp = C.POINTER(C.c_double)

    def arr(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(dp)

    (o, op), (a1, e1p), (a2, e2p), (x, xp), (z, zp) = arr(origin), arr(e1), arr(e2), arr(xs), arr(zs)
    n = len(x)
    if len(z) != n:
        raise ValueError("xs and zs must have the same length")
    if hits_device_ptr is None:
        h = np.ascontiguousarray(np.asarray(hits, dtype=np.float64).reshape(-1, 9))
        hp, nh, on_dev = h.ctypes.data_as(C.c_void_p), len(h), 0
    else:
        hp, nh, on_dev = C.c_void_p(int(hits_device_ptr)), int(n_hits), 1
    out = np.zeros(n * n)
    fld = np.zeros(2 * n * n) if want_field else None
    ms = C.c_double()
    check(lib, lib.bmo_psf_intensity(hp, nh, on_dev, op, e1p, e2p, xp, zp, n, int(device), out.ctypes.data_as(dp),
                                     fld.ctypes.data_as(dp) if want_field else None, C.byref(ms)), "bmo_psf_intensity")
    field = (fld[0::2] + 1j * fld[1::2]).reshape(n, n).T.copy() if want_field else None
    return out.reshape(n, n).T.copy(), field, ms.value
