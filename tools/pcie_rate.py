"""PCIe-inclusive rates of the host-buffer boundary (bmo_trace) on config C2, beside the HBM-resident rate bench.py reports.

  resident      bmo_trace_device: batch already in HBM, result left in HBM (bench.py's `value`)
  upload+solve  bmo_trace: host ray batch in, result left in HBM
  +hits         ... plus the detector hit tables copied to host memory (what a Spotdetector user reads)
  +full log     ... plus bmo_result_view: the whole segment log, node table and hits copied to host memory
"""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bmo_amd as bmo
from bmo_amd import abi
from bmo_amd.system import make_batch
import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
system = scenes.c2_scene()[0]
bundle = scenes.c2_bundle(n)
scene = bmo.CompiledScene(system, bundle.lambdas)
eng = bmo.Engine(scene, 0)
lib = eng.lib
hip = C.CDLL('/opt/rocm/lib/libamdhip64.so')
batch, keep = make_batch(scene, bundle)
o = eng.opts(100)


def best_of(fn, reps=4):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


dev = eng.upload(bundle)
hit_bufs = {}
def resident():
    res = eng.trace_device(dev, 100)
    eng.free_result(res)
def host_in():
    res = C.c_void_p()
    abi.check(lib, lib.bmo_trace(eng.handle, C.byref(batch), C.byref(o), C.byref(res)), "bmo_trace")
    lib.bmo_result_free(res)
def host_in_hits():
    res = C.c_void_p()
    abi.check(lib, lib.bmo_trace(eng.handle, C.byref(batch), C.byref(o), C.byref(res)), "bmo_trace")
    tot = 0
    for s in range(len(scene.detectors)):
        p, cnt = eng.result_device_hits(res, s)
        if s not in hit_bufs or hit_bufs[s].shape[0] < cnt:
            hit_bufs[s] = np.zeros((cnt, 9))  # the caller's buffer, reused across solves
        if cnt:
            eng.result_copy_hits(res, s, hit_bufs[s].ctypes.data, cnt)
        tot += cnt
    lib.bmo_result_free(res)
    return tot
def host_in_full():
    res = C.c_void_p()
    abi.check(lib, lib.bmo_trace(eng.handle, C.byref(batch), C.byref(o), C.byref(res)), "bmo_trace")
    v = abi.ResultView()
    abi.check(lib, lib.bmo_result_view(res, C.byref(v)), "bmo_result_view")
    out = (v.n_intersect_calls, v.n_records)
    lib.bmo_result_free(res)
    return out

resident()
t_res, _ = best_of(resident)
t_in, _ = best_of(host_in)
t_hits, nh = best_of(host_in_hits)
t_full, (calls, recs) = best_of(host_in_full, 2)
for name, t in (("resident", t_res), ("upload+solve", t_in), ("+hits to host", t_hits), ("+full log to host", t_full)):
    print("%-18s %9.3f ms  %.3e intersections/s" % (name, t * 1e3, calls / t), flush=True)
print("rays %d  intersect3d calls %d  segment records %d  hits %d" % (n, calls, recs, nh))
