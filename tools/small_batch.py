"""Latency of small solves (the reference's interactive sizes): C1 and C2 scenes at 100 .. 32768 rays, wall time per bmo_trace_device."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes
for name, system, mk in (("C1", scenes.c1_scene()[0], scenes.c1_bundle), ("C2", scenes.c2_scene()[0], scenes.c2_bundle)):
    for n in (100, 1000, 10000, 32768, 65536):
        bundle = mk(n)
        scene = bmo.CompiledScene(system, bundle.lambdas)
        eng = bmo.Engine(scene, 0)
        dev = eng.upload(bundle)
        best = None
        for rep in range(6):
            t = time.perf_counter()
            res = eng.trace_device(dev, 100)
            dt = time.perf_counter() - t
            k, tot, nl = eng.result_timing(res)
            eng.free_result(res)
            if best is None or dt < best[0]:
                best = (dt, k, nl)
        eng.free_batch(dev); eng.close()
        print("%s n=%6d  wall %.3f ms  kernels %.3f ms  launches %d" % (name, n, best[0] * 1e3, best[1], best[2]), flush=True)
