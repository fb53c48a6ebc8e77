"""Retrace through the host-batch form of the boundary (bmo_trace, then bmo_retrace with the root heads uploaded again on every call — what
solve_system of the Python / Julia wrappers does) on config 2 under SURVEY 8(d)'s bundle: kernel time of the first solve and of three retraces
after a 0.05 deg tilt of the splitter.  The retraces order their tiles by the times the retraced solution recorded (bmo_trace_result::tile_cost)."""
import ctypes as C, math, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes
from bmo_amd import abi
from test_retrace import _c2_with_parts

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
system, parts = _c2_with_parts()
bundle = scenes.c2_survey_bundle(n)
scene0 = bmo.CompiledScene(system, bundle.lambdas)
bmo.xrotate3d(parts["bs"], math.radians(0.05))
scene1 = bmo.CompiledScene(system, bundle.lambdas)
e0, e1 = bmo.Engine(scene0, 0), bmo.Engine(scene1, 0)
lib = e0.lib
batch, keep = bmo.make_batch(scene0, bundle)
o = e0.opts(20)
prev = C.c_void_p()
abi.check(lib, lib.bmo_trace(e0.handle, C.byref(batch), C.byref(o), C.byref(prev)), "bmo_trace")
print("first solve: kernels %.3f ms" % e0.result_timing(prev)[0], flush=True)
for rep in range(3):
    re = C.c_void_p()
    abi.check(lib, lib.bmo_retrace(e1.handle, C.byref(batch), prev, C.byref(o), C.byref(re)), "bmo_retrace")
    k, t, nl = e1.result_timing(re)
    print("retrace %d: kernels %.3f ms, %d launches" % (rep, k, nl), flush=True)
    lib.bmo_result_free(re)
