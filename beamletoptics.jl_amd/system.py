"""System container, scene compiler and the `solve_system` entry point.

Mirrors src/System.jl: `System(objects)` (:10-21) and
`solve_system!(system, beam|group; r_max=100)` (:444-468).  The scene is flattened in
`Leaves` order into the tables of include/bmo.h and handed to the HIP engine through the
C ABI; the beam trees are rebuilt from the engine's result tables in reference order.
"""
import ctypes as C

import numpy as np

from . import abi
from . import beams as bm
from . import components as cp
from . import shapes as sh

# miss-cull inflation of bounding spheres (see DESIGN.md "miss cull")
_BS_REL, _BS_ABS = 1e-6, 1e-6


class System:
    """src/System.jl:10-21"""

    def __init__(self, objects):
        self.objects_tree = list(objects) if isinstance(objects, (list, tuple)) else [objects]

    def objects(self):
        return cp.leaves(self.objects_tree)


StaticSystem = System  # src/System.jl:38-45: same tracing semantics


class CompiledScene:
    """Flat tables + the ctypes descriptor that points into them (kept alive together)."""

    def __init__(self, system, lambdas, cull=True, consts=None):
        objs = system.objects() if isinstance(system, System) else cp.leaves(system)
        self.leaf_objects = objs
        self.lambdas = np.array(sorted(set(float(l) for l in lambdas)), dtype=np.float64)
        self.shape_list = []   # python shape per shape id
        self._shape_ids = {}
        shapes, children, tris, coefs = [], [], [], []
        media, media_ids = [], {}

        def medium_of(fn):
            key = id(fn)
            if key not in media_ids:
                media_ids[key] = len(media)
                media.append([float(fn(l)) for l in self.lambdas])
            return media_ids[key]

        slope = {}  # shape id -> K of "sdf >= dist / K" (1 for exact sdfs; None: no bound, never culled)

        def add_shape(s):
            if id(s) in self._shape_ids:
                return self._shape_ids[id(s)]
            rec = abi.Shape()
            sid = len(shapes)
            shapes.append(rec)
            self.shape_list.append(s)
            self._shape_ids[id(s)] = sid
            rec.kind = s.kind
            rec.child_begin = rec.child_count = rec.tri_begin = rec.tri_count = rec.flags = 0
            rec.pos[:] = list(s.pos)
            rec.dir[:] = list(np.asarray(s.dir, dtype=np.float64).reshape(9))
            tdir = s.tdir if hasattr(s, "tdir") else np.asarray(s.dir).T
            rec.tdir[:] = list(np.asarray(tdir, dtype=np.float64).reshape(9))
            p = list(s.params()) if hasattr(s, "params") else []
            rec.p[:] = p + [0.0] * (abi.NPARAM - len(p))
            if s.kind == sh.K_MESH:
                t = s.triangles()
                rec.tri_begin = len(tris)
                rec.tri_count = len(t)
                tris.extend(t)
                v = s.vertices
                c = (v.min(axis=0) + v.max(axis=0)) / 2
                r = float(np.sqrt(((v - c) ** 2).sum(axis=1)).max())
            elif s.kind in (sh.K_UNION, sh.K_MENISCUS):
                ids = [add_shape(ch) for ch in s.sdfs]
                rec.child_begin = len(children)
                rec.child_count = len(ids)
                children.extend(ids)
                c, r = s.world_bound()
                if any(shapes[i].flags & sh.FLAG_INEXACT for i in ids):
                    rec.flags |= sh.FLAG_INEXACT
                ks = [slope[i] for i in (ids[:2] if s.kind == sh.K_MENISCUS else ids)]  # min over children: the weakest bound holds
                slope[sid] = None if any(k is None for k in ks) else max(ks)
            elif s.kind in (sh.K_ASPH_CONVEX, sh.K_ASPH_CONCAVE, sh.K_ACYL_CONVEX, sh.K_ACYL_CONCAVE):
                rec.child_begin = len(coefs)
                rec.child_count = len(s.coefficients)
                coefs.extend(s.coefficients)
                rec.flags |= sh.FLAG_INEXACT
                c, r = s.world_bound()
                slope[sid] = s.slope_bound()
            else:
                c, r = s.world_bound()
            K = slope.setdefault(sid, 1.0)
            if cull and K is not None:
                # outside the inflated sphere dist(p, solid) >= K * (r * 1e-6 + 1 um), hence sdf(p) >= r * 1e-6 + 1 um: the
                # reference's start classification is "outside" (> 1e-9) and its hit test (< 1e-10) can never fire
                rec.bs_center[:] = list(c)
                rec.bs_radius = r + K * (r * _BS_REL + _BS_ABS)
            else:
                rec.bs_center[:] = [0.0, 0.0, 0.0]
                rec.bs_radius = -1.0
            return sid

        objects = []
        self.detectors = []
        for o in objs:
            rec = abi.Object()
            rec.kind = o.kind
            rec.shape[:] = [-1, -1, -1]
            rec.medium[:] = [-1, -1]
            rec.detector = -1
            rec.reflectance = rec.transmittance = rec.cutoff = 0.0
            parts = o.parts()
            for k, s in enumerate(parts):
                rec.shape[k] = add_shape(s)
            if o.kind == cp.O_REFRACTIVE:
                rec.medium[0] = medium_of(o.n)
            elif o.kind == cp.O_DOUBLET:
                rec.medium[0] = medium_of(o.front.n)
                rec.medium[1] = medium_of(o.back.n)
            elif o.kind == cp.O_THIN_BS:
                rec.reflectance, rec.transmittance = o.reflectance, o.transmittance
            elif o.kind == cp.O_PLATE_BS:
                rec.medium[0] = medium_of(o.substrate.n)
                rec.reflectance, rec.transmittance = o.coating.reflectance, o.coating.transmittance
            elif o.kind == cp.O_CUBE_BS:
                rec.medium[0] = medium_of(o.front.n)
                rec.medium[1] = medium_of(o.back.n)
                rec.reflectance, rec.transmittance = o.coating.reflectance, o.coating.transmittance
            elif o.kind in (cp.O_SPOT, cp.O_PSF, cp.O_PHOTODETECTOR):
                rec.detector = len(self.detectors)
                self.detectors.append(o)
            elif o.kind == cp.O_POLARIZER:
                rec.cutoff = o.cutoff
                j = np.asarray(o.jones, dtype=np.complex128).reshape(9)
                rec.jones[:] = [x for z in j for x in (z.real, z.imag)]
            objects.append(rec)

        self.n_objects = len(objects)
        self._objects = (abi.Object * max(1, len(objects)))(*objects)
        self._shapes = (abi.Shape * max(1, len(shapes)))(*shapes)
        self._children = np.array(children if children else [0], dtype=np.int32)
        self._tris = np.array(tris if tris else [[0.0] * 9], dtype=np.float64).reshape(-1)
        self._ntab = np.array(media if media else [[1.0] * max(1, len(self.lambdas))], dtype=np.float64).reshape(-1)
        self._coefs = np.array(coefs if coefs else [0.0], dtype=np.float64)
        d = abi.SceneDesc()
        d.abi_version = abi.ABI_VERSION
        d.n_objects, d.n_shapes, d.n_children = len(objects), len(shapes), len(children)
        d.n_tris, d.n_media, d.n_lambda, d.n_detectors = len(tris), len(media), len(self.lambdas), len(self.detectors)
        d.objects = C.cast(self._objects, C.POINTER(abi.Object))
        d.shapes = C.cast(self._shapes, C.POINTER(abi.Shape))
        d.children = self._children.ctypes.data_as(C.POINTER(C.c_int32))
        d.tris = self._tris.ctypes.data_as(C.POINTER(C.c_double))
        d.n_table = self._ntab.ctypes.data_as(C.POINTER(C.c_double))
        d.lambdas = self.lambdas.ctypes.data_as(C.POINTER(C.c_double))
        d.coefs = self._coefs.ctypes.data_as(C.POINTER(C.c_double))
        d.n_coefs = len(coefs)
        k = dict(eps_srf=1e-9, eps_ray=1e-10, eps_ins=1.0, mt_keps=1e-9, mt_leps=1e-9, grad_h=1e-8, march_iters=1000)
        k.update(consts or {})
        for name, val in k.items():
            setattr(d, name, val)
        self.desc = d

    def shape_id(self, shape):
        return self._shape_ids[id(shape)]

    def lambda_index(self, lams):
        lams = np.asarray(lams, dtype=np.float64)
        idx = np.searchsorted(self.lambdas, lams)
        if np.any(idx >= len(self.lambdas)) or np.any(self.lambdas[np.minimum(idx, len(self.lambdas) - 1)] != lams):
            raise KeyError("wavelength not in the compiled lambda table")
        return idx.astype(np.int32)


def make_batch(scene, bundle):
    """bmo_ray_batch for a RayBundle; returns (batch, keepalive)."""
    li = scene.lambda_index(bundle.lambdas)
    planes = np.ascontiguousarray(bundle.planes, dtype=np.float64)
    b = abi.RayBatch()
    b.n = bundle.n
    b.kind = bundle.kind
    b.n_planes = planes.shape[0]
    b.planes = planes.ctypes.data_as(C.POINTER(C.c_double))
    b.lambda_idx = li.ctypes.data_as(C.POINTER(C.c_int32))
    return b, (planes, li)


class Engine:
    """Thin RAII wrapper over the C ABI for one compiled scene."""

    def __init__(self, scene, device=0, max_beams=0):
        self.lib = abi.load_engine()
        self.scene = scene
        self.device = device
        self.max_beams = max_beams  # bmo_trace_opts.max_beams: 0 = no limit (reference behaviour)
        self.handle = C.c_void_p()
        abi.check(self.lib, self.lib.bmo_scene_create(C.byref(scene.desc), C.byref(self.handle)), "bmo_scene_create")

    def close(self):
        if self.handle:
            self.lib.bmo_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def opts(self, r_max, record_segments=True):
        o = abi.TraceOpts()
        o.r_max, o.device, o.record_segments, o.max_beams = int(r_max), int(self.device), int(bool(record_segments)), int(self.max_beams)
        return o

    def trace(self, bundle, r_max=100):
        """bmo_trace: upload + trace + download; returns abi.TraceResult."""
        batch, keep = make_batch(self.scene, bundle)
        res = C.c_void_p()
        o = self.opts(r_max)
        abi.check(self.lib, self.lib.bmo_trace(self.handle, C.byref(batch), C.byref(o), C.byref(res)), "bmo_trace")
        try:
            v = abi.ResultView()
            abi.check(self.lib, self.lib.bmo_result_view(res, C.byref(v)), "bmo_result_view")
            out = abi.TraceResult(v)
            ms, tot, nl = C.c_double(), C.c_double(), C.c_int32()
            self.lib.bmo_result_timing(res, C.byref(ms), C.byref(tot), C.byref(nl))
            out.kernel_ms, out.total_ms, out.n_launches = ms.value, tot.value, nl.value
        finally:
            self.lib.bmo_result_free(res)
        return out

    # split form (inputs resident in HBM)
    def upload(self, bundle):
        batch, keep = make_batch(self.scene, bundle)
        h = C.c_void_p()
        abi.check(self.lib, self.lib.bmo_batch_upload(self.handle, C.byref(batch), self.device, C.byref(h)), "bmo_batch_upload")
        return h

    def free_batch(self, h):
        self.lib.bmo_batch_free(h)

    def trace_device(self, dev_batch, r_max=100, record_segments=True):
        res = C.c_void_p()
        o = self.opts(r_max, record_segments)
        abi.check(self.lib, self.lib.bmo_trace_device(self.handle, dev_batch, C.byref(o), C.byref(res)), "bmo_trace_device")
        return res

    def retrace_device(self, dev_batch, prev, r_max=100):
        """bmo_retrace_device: re-solve the batch of `prev` (a result handle) against this engine's scene."""
        res = C.c_void_p()
        o = self.opts(r_max)
        abi.check(self.lib, self.lib.bmo_retrace_device(self.handle, dev_batch, prev, C.byref(o), C.byref(res)), "bmo_retrace_device")
        return res

    def result_timing(self, res):
        ms, tot, nl = C.c_double(), C.c_double(), C.c_int32()
        abi.check(self.lib, self.lib.bmo_result_timing(res, C.byref(ms), C.byref(tot), C.byref(nl)), "bmo_result_timing")
        return ms.value, tot.value, nl.value

    def result_view(self, res, what=None):
        """bmo_result_view (the whole solution) or, with `what` (abi.VIEW_* mask), bmo_result_view_select."""
        v = abi.ResultView()
        if what is None:
            abi.check(self.lib, self.lib.bmo_result_view(res, C.byref(v)), "bmo_result_view")
        else:
            abi.check(self.lib, self.lib.bmo_result_view_select(res, int(what), C.byref(v)), "bmo_result_view_select")
        return abi.TraceResult(v)

    def result_device_hits(self, res, slot):
        p = C.POINTER(C.c_double)()
        n = C.c_int64()
        abi.check(self.lib, self.lib.bmo_result_device_hits(res, slot, C.byref(p), C.byref(n)), "bmo_result_device_hits")
        return C.cast(p, C.c_void_p).value or 0, n.value

    def result_copy_hits(self, res, slot, dst_device_ptr, max_hits):
        abi.check(self.lib, self.lib.bmo_result_copy_hits(res, slot, C.c_void_p(dst_device_ptr), max_hits), "bmo_result_copy_hits")

    def result_copy_hit_columns(self, res, slot, n_cols, dst_ptr, max_hits):
        """The leading n_cols columns of the ordered hit table of one detector, packed, into device or host memory."""
        abi.check(self.lib, self.lib.bmo_result_copy_hit_columns(res, slot, int(n_cols), C.c_void_p(dst_ptr), max_hits), "bmo_result_copy_hit_columns")

    def result_counts(self, res):
        """Per-detector hit counts without downloading anything."""
        return [self.result_device_hits(res, slot)[1] for slot in range(len(self.scene.detectors))]

    def result_size(self, res):
        """bmo_result_counts: (reference intersect3d calls, segments, beams, detector hits) without downloading the solution."""
        c, r, n, h = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        abi.check(self.lib, self.lib.bmo_result_counts(res, C.byref(c), C.byref(r), C.byref(n), C.byref(h)), "bmo_result_counts")
        return c.value, r.value, n.value, h.value

    def free_result(self, res):
        self.lib.bmo_result_free(res)


# ----------------------------------------------------------------------------------------
def _fill_beams(scene, res, roots):
    """Rebuild Beam / GaussianBeamlet trees in reference order from a TraceResult."""
    objs = scene.leaf_objects
    shapes = scene.shape_list
    kind = res.beam_kind
    nodes = [None] * res.n_nodes

    def make_rays(node, base, first):
        rays = []
        f, n = int(res.node_first_rec[node]), int(res.node_nseg[node])
        lam = res.node_aux[node, 3] if kind == bm.BEAM_GAUSSIAN else res.node_aux[node, 0]
        for r in range(f, f + n):
            P = res.rec[:, r]
            if kind == bm.BEAM_POLARIZED:
                ray = bm.PolarizedRay.__new__(bm.PolarizedRay)
                ray.E0 = P[11:17:2] + 1j * P[12:17:2]
            else:
                ray = bm.Ray.__new__(bm.Ray)
            ray.pos = P[base:base + 3].copy()
            ray.dir = P[base + 3:base + 6].copy()
            ray.n = float(P[base + 6])
            ray.lam = float(lam)
            t = float(P[base + 7])
            if np.isfinite(t):
                o, s = int(res.rec_obj[r]), int(res.rec_shape[r])
                ray.intersection = bm.Intersection(t, P[base + 8:base + 11].copy(), objs[o] if o >= 0 else None,
                                                   shapes[s] if s >= 0 else None)
            else:
                ray.intersection = None
            rays.append(ray)
        return rays

    for i in range(res.n_nodes):
        par = int(res.node_parent[i])
        if par < 0:
            b = roots[int(res.node_root[i])]
        elif kind == bm.BEAM_GAUSSIAN:
            b = bm.GaussianBeamlet(None, None, _parts=(bm.Beam.__new__(bm.Beam), bm.Beam.__new__(bm.Beam), bm.Beam.__new__(bm.Beam),
                                                       0.0, 0.0, 0j))
            for sub in (b.chief, b.waist, b.divergence):
                sub.parent, sub.children, sub.status = None, [], 0
        else:
            b = bm.Beam.__new__(bm.Beam)
            b.parent, b.children, b.status = None, [], 0
        nodes[i] = b
        b.status = int(res.node_status[i])
        b.children = []
        if kind == bm.BEAM_GAUSSIAN:
            b.chief.rays = make_rays(i, 0, True)
            b.waist.rays = make_rays(i, 11, False)
            b.divergence.rays = make_rays(i, 22, False)
            b.w0 = float(res.node_aux[i, 0])
            b.E0 = complex(res.node_aux[i, 1], res.node_aux[i, 2])
            b.lam = float(res.node_aux[i, 3])
        else:
            b.rays = make_rays(i, 0, True)
        if par >= 0:
            p = nodes[par]
            b.parent = p
            p.children.append(b)
            if kind == bm.BEAM_GAUSSIAN:
                b.chief.parent = p.chief  # parent! Gaussian.jl:113-117
    return nodes


class EngineSolution:
    """A solved batch that stays resident on the GPU (bmo_trace_result*), so that the next solve_system can retrace it."""

    def __init__(self, lib, handle, n_roots, kind):
        self.lib, self.handle, self.n_roots, self.kind = lib, handle, n_roots, kind

    def free(self):
        if self.handle:
            self.lib.bmo_result_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def gauss_parameters(self, node, zs):
        """bmo_gauss_parameters: rows (w, R, psi, w0) of beamlet `node` (result order) at the distances `zs` along the beam."""
        dp = C.POINTER(C.c_double)
        z = np.ascontiguousarray(zs, dtype=np.float64)
        out = np.zeros((len(z), 4))
        abi.check(self.lib, self.lib.bmo_gauss_parameters(self.handle, int(node), z.ctypes.data_as(dp), len(z), out.ctypes.data_as(dp)), "bmo_gauss_parameters")
        return out

    def set_gauss_prefix(self, prefix_start, prefix_segs, opl_parent):
        """bmo_result_set_gauss_prefix: the earlier segments of continued beamlets ([n_roots + 1] starts, [24, total] planes, [n_roots] parent OPL)."""
        st = np.ascontiguousarray(prefix_start, dtype=np.int32)
        sg = np.ascontiguousarray(prefix_segs, dtype=np.float64)
        op = np.ascontiguousarray(opl_parent, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        abi.check(self.lib, self.lib.bmo_result_set_gauss_prefix(self.handle, len(op), st.ctypes.data_as(C.POINTER(C.c_int32)), sg.ctypes.data_as(dp),
                                                                 op.ctypes.data_as(dp)), "bmo_result_set_gauss_prefix")

    def photodetector_field(self, slot, position, orientation, xs, ys, field):
        """bmo_photodetector_field: adds the field of the beamlets recorded on detector `slot` to `field[i, j]` (in place)."""
        dp = C.POINTER(C.c_double)
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (position, np.asarray(orientation).reshape(9), xs, ys)]
        nx, ny = len(a[2]), len(a[3])
        buf = np.zeros(2 * nx * ny)  # (i, j) at [i + nx*j]
        ft = np.ascontiguousarray(field.T)
        buf[0::2], buf[1::2] = ft.real.reshape(-1), ft.imag.reshape(-1)
        ms = C.c_double()
        abi.check(self.lib, self.lib.bmo_photodetector_field(self.handle, int(slot), a[0].ctypes.data_as(dp), a[1].ctypes.data_as(dp),
                                                             a[2].ctypes.data_as(dp), a[3].ctypes.data_as(dp), nx, ny, buf.ctypes.data_as(dp),
                                                             C.byref(ms)), "bmo_photodetector_field")
        field[...] = (buf[0::2] + 1j * buf[1::2]).reshape(ny, nx).T
        return ms.value


def _engine_solve(scene, bundle, r_max, prev, device=0, max_beams=0):
    """One solve on the HIP engine: bmo_trace, or bmo_retrace when `prev` (an EngineSolution) is given."""
    eng = Engine(scene, device, max_beams)
    try:
        batch, keep = make_batch(scene, bundle)
        h = C.c_void_p()
        o = eng.opts(r_max)
        if prev is None:
            abi.check(eng.lib, eng.lib.bmo_trace(eng.handle, C.byref(batch), C.byref(o), C.byref(h)), "bmo_trace")
        else:
            abi.check(eng.lib, eng.lib.bmo_retrace(eng.handle, C.byref(batch), prev.handle, C.byref(o), C.byref(h)), "bmo_retrace")
        sol = EngineSolution(eng.lib, h, bundle.n, bundle.kind)
        res = eng.result_view(h)
    finally:
        eng.close()
    return res, sol


def release(beams):
    """Free the solution kept for retracing (device memory of the segment log) of a solved beam / beam group."""
    roots = beams.beams if isinstance(beams, bm.BeamGroup) else (list(beams) if isinstance(beams, (list, tuple)) else [beams])
    for b in roots:
        sol = getattr(b, "_solution", None)
        if sol is not None:
            sol.free()
        b._solution = None


def _roots_of(beams):
    if isinstance(beams, bm.BeamGroup):
        return beams.beams
    if isinstance(beams, (list, tuple)):
        return list(beams)
    return [beams]


def _prepare(system, beams):
    """(roots, bundle of their first rays, compiled scene, solution of the previous solve of exactly these beams or None)."""
    roots = _roots_of(beams)
    bundle = bm.RayBundle.from_beams(roots)
    scene = CompiledScene(system, bundle.lambdas)
    prev = getattr(roots[0], "_solution", None) if roots else None
    if prev is not None:
        same = all(getattr(b, "_solution", None) is prev and getattr(b, "_sol_index", -1) == i for i, b in enumerate(roots))
        if not (same and prev.handle and prev.n_roots == len(roots) and prev.kind == bundle.kind):
            prev = None  # a different grouping than the one that was solved: treat as fresh beams
    return roots, bundle, scene, prev


def _apply(scene, res, sol, roots, hit_fix=None, defer=False):
    """Fill the beam objects from a TraceResult and append this solve's detector records (detectors are not reset, like the reference:
    Spotdetector.jl / PSFDetector.jl "Reset behavior").  `hit_fix(slot, hits)` may adjust the rows before they are appended.  With `defer`
    the Spot / PSF rows are not appended but left on the beam that made them (`_pending_hits`), for a caller that merges several solves into
    the reference's order (_trace_open_leaves)."""
    nodes = _fill_beams(scene, res, roots)
    for i, b in enumerate(roots):
        b._solution, b._sol_index = sol, i
    for slot, det in enumerate(scene.detectors):
        hits = res.detector_hits(slot)
        if hit_fix is not None:
            hits = hit_fix(slot, hits)
        if defer and det.kind != cp.O_PHOTODETECTOR:
            rows = hits[:, 0:2] if det.kind == cp.O_SPOT else hits
            for nd, row in zip(res.detector_nodes(slot), rows):
                nodes[int(nd)].__dict__.setdefault("_pending_hits", []).append((det, row))
            continue
        if det.kind == cp.O_SPOT:
            det.data = np.concatenate([det.data, hits[:, 0:2]], axis=0)
        elif det.kind == cp.O_PHOTODETECTOR:
            if len(hits):  # pd.field[i, j] += electric_field(gauss, r, z) * sqrt(proj) for every recorded beamlet (Photodetector.jl:85-105)
                sol.photodetector_field(slot, det.position(), det.orientation(), det.x, det.y, det.field)
        else:
            det.data = np.concatenate([det.data, hits], axis=0)
    return nodes


def _trace_open_leaves(system, roots, r_max, device):
    """solve_system!(...; retrace = false) on beams that were solved before (System.jl:449-458, solve_leaf! :470-475): nothing is re-walked;
    every beam of the trees whose LAST ray has no intersection is traced on from that ray (without a hint, up to r_max rays in the beam),
    everything else stays as it is.  The open last rays are traced as one fresh batch per distinct remaining length and spliced back.
    A batch applies ONE ray limit to all its beams, the remaining length of the beams it continues, while the reference gives every child
    born in the continuation the full r_max: children that ran into the batch's limit are open beams themselves and are continued by the next
    pass.  Detector rows are appended at the end, in the order the reference's loop makes them: root by root, breadth-first."""
    open_beams, queue = [], list(roots)
    while queue:  # BFS like solve_system!
        b = queue.pop(0)
        queue.extend(b.children)
        if b.rays[-1].intersection is None and len(b.rays) < r_max:
            open_beams.append(b)
    while open_beams:
        open_beams = _continue_open_beams(system, open_beams, r_max, device)
    for root in roots:  # solve_system!(bg) finishes root 1's whole tree before it starts root 2 (System.jl:463-468): one BFS per root
        queue = [root]
        while queue:
            b = queue.pop(0)
            queue.extend(b.children)
            for det, row in b.__dict__.pop("_pending_hits", []):
                det.data = np.concatenate([det.data, np.asarray(row)[None, :]], axis=0)
    for b in roots:  # the resident solution no longer describes these beams: the next solve traces them afresh
        sol = getattr(b, "_solution", None)
        if sol is not None:
            sol.free()
        b._solution = None
    return None


def _continue_open_beams(system, open_beams, r_max, device):
    """One pass of _trace_open_leaves; returns the beams born in it that ran into the pass's ray limit below r_max."""
    again = []
    by_left = {}
    for b in open_beams:
        by_left.setdefault(r_max - len(b.rays) + 1, []).append(b)
    for left, group in sorted(by_left.items()):
        gaussian = group[0].kind == bm.BEAM_GAUSSIAN
        heads = []
        for b in group:
            if gaussian:
                subs = []
                for part in (b.chief, b.waist, b.divergence):
                    sub = bm.Beam.__new__(bm.Beam)
                    sub.rays, sub.parent, sub.children, sub.status = [part.rays[-1]], None, [], 0
                    subs.append(sub)
                h = bm.GaussianBeamlet(None, None, _parts=(subs[0], subs[1], subs[2], b.lam, b.w0, b.E0))
                h.parent, h.children, h.status = None, [], 0
            else:
                h = bm.Beam.__new__(bm.Beam)
                h.rays, h.parent, h.children, h.status = [b.rays[-1]], None, [], 0
            heads.append(h)
        bundle = bm.RayBundle.from_beams(heads)
        if gaussian:
            # the lengths the beamlets have accumulated up to their open rays (bmo.h "31 planes"), folded exactly as a solve that had
            # never stopped would have folded them
            acc = np.zeros((6, len(group)))
            for i, b in enumerate(group):
                l0 = 0.0 if b.parent is None else b.parent.length()
                len_a, len_b = 0.0, l0
                opl_c = 0.0 if b.chief.parent is None else b.chief.parent.optical_path_length()
                for r in b.chief.rays[:-1]:
                    len_a += r.intersection.t
                    len_b += r.intersection.t
                    opl_c += r.intersection.t * r.n
                opl_w = opl_d = 0.0
                for r in b.waist.rays[:-1]:
                    opl_w += r.intersection.t * r.n
                for r in b.divergence.rays[:-1]:
                    opl_d += r.intersection.t * r.n
                acc[:, i] = (len_a, len_b, l0, opl_c, opl_w, opl_d)
            bundle = bm.RayBundle(bundle.kind, np.vstack([bundle.planes, acc]))
        scene = CompiledScene(system, bundle.lambdas)
        res, sol = _engine_solve(scene, bundle, left, None, device)
        if gaussian and any(isinstance(o, cp.Photodetector) for o in system.objects()):
            # the field of a beamlet on a Photodetector is a function of ALL its rays (point_on_beam, length, optical_path_length:
            # Beam.jl:125-205); the continuation holds only those from the open ray on, the ones in front of it go along as a prefix
            starts, cols, opl_par = [0], [], []
            for b in group:
                for k in range(len(b.chief.rays) - 1):
                    col = []
                    for part in (b.chief, b.waist, b.divergence):
                        r = part.rays[k]
                        col += [r.pos[0], r.pos[1], r.pos[2], r.dir[0], r.dir[1], r.dir[2], r.n, r.intersection.t]
                    cols.append(col)
                starts.append(len(cols))
                opl_par.append(0.0 if b.chief.parent is None else b.chief.parent.optical_path_length())
            segs = np.asarray(cols, dtype=np.float64).T if cols else np.zeros((24, 0))
            sol.set_gauss_prefix(starts, segs, opl_par)
        # optical path length of each beam (parents included) up to the start of its open ray: what PSF records of the continuation add
        opl0 = []
        for b in group:
            acc = 0.0 if b.parent is None else b.parent.optical_path_length()
            for r in b.rays[:-1]:
                acc += r.optical_path_length()
            opl0.append(acc)

        def hit_fix(slot, hits, res=res, scene=scene, opl0=opl0, gaussian=gaussian):
            if gaussian or scene.detectors[slot].kind != cp.O_PSF or not len(hits):  # (a continued beamlet brings its path lengths along)
                return hits
            hits = hits.copy()
            root_of_hit = res.node_root[res.detector_nodes(slot)]
            hits[:, 6] += np.asarray(opl0)[root_of_hit]
            return hits

        nodes = _apply(scene, res, sol, heads, hit_fix, defer=True)
        sol.free()
        for b, h in zip(group, heads):
            if gaussian:
                for part, hp in ((b.chief, h.chief), (b.waist, h.waist), (b.divergence, h.divergence)):
                    part.rays[-1:] = hp.rays
                for c in h.children:
                    c.chief.parent = b.chief  # parent! Gaussian.jl:113-117
            else:
                b.rays[-1:] = h.rays  # the open ray (now with its intersection, if any) and what followed it
            b.status = h.status
            for c in h.children:
                c.parent = b
            b.children = h.children
            pend = h.__dict__.pop("_pending_hits", None)
            if pend:
                b.__dict__.setdefault("_pending_hits", []).extend(pend)
        if left < r_max:
            again += [nd for nd in nodes[len(group):] if (nd.status & abi.NODE_RMAX) and nd.rays[-1].intersection is None and len(nd.rays) < r_max]
    return again


def solve_system(system, beams, r_max=100, retrace=True, device=0):
    """solve_system!(system, beam | beam group; r_max=100, retrace=true) — src/System.jl:444-468.

    Fresh beams are traced; beams solved by an earlier call are RETRACED (System.jl:188-255, :326-428): the stored path is
    re-walked against the system as it is now, the first ray of every root supplying the (possibly modified) head.
    With retrace=False solved beams are not re-walked: only leaves whose last ray has no intersection are traced on (System.jl:470-475).
    Mutates the beam objects (rays, intersections, children) and appends detector data, exactly like the reference
    (detectors are not reset, Spotdetector.jl / PSFDetector.jl "Reset behavior").  Returns the raw TraceResult (None for retrace=False
    on solved beams).  The backend is the HIP engine, always: there is no other.
    """
    roots, bundle, scene, prev = _prepare(system, beams)
    if prev is not None and not retrace:
        return _trace_open_leaves(system, roots, r_max, device)
    res, sol = _engine_solve(scene, bundle, r_max, prev, device)
    _apply(scene, res, sol, roots)
    return res
