"""Retrace vs fresh trace on C2 (2^20 rays): solve, tilt the splitter by 0.05 deg, re-solve with bmo_retrace_device."""
import math, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes
from test_retrace import _c2_with_parts

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
system, parts = _c2_with_parts()
bundle = scenes.c2_bundle(n)
scene0 = bmo.CompiledScene(system, bundle.lambdas)
bmo.xrotate3d(parts["bs"], math.radians(0.05))
scene1 = bmo.CompiledScene(system, bundle.lambdas)
e0, e1 = bmo.Engine(scene0, 0), bmo.Engine(scene1, 0)
dev = e0.upload(bundle)
prev = e0.trace_device(dev, 20)
for rep in range(3):
    fresh = e1.trace_device(dev, 20)
    kf, tf, nf = e1.result_timing(fresh)
    vf = None
    re = e1.retrace_device(dev, prev, 20)
    kr, tr, nr = e1.result_timing(re)
    print("fresh   kernels %8.3f ms total %8.3f ms launches %d | retrace kernels %8.3f ms total %8.3f ms launches %d" % (kf, tf, nf, kr, tr, nr), flush=True)
    if rep == 2:
        a, b = e1.result_view(fresh), e1.result_view(re)
        print("intersect3d calls fresh %d retrace %d ; records %d %d ; nodes %d %d" % (a.n_intersect_calls, b.n_intersect_calls, a.n_records, b.n_records, a.n_nodes, b.n_nodes))
    e1.free_result(fresh); e1.free_result(re)
