"""record_segments = 0 (detector-only solves): time against the full-log solve on C2, and a bundle too large to log comfortably."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes
system = scenes.c2_scene()[0]
sizes = ((1 << 20, (True, False)), (1 << 24, (False,))) if len(sys.argv) < 2 else ((1 << int(sys.argv[1]), (False,)),)
for n, modes in sizes:
    bundle = scenes.c2_bundle(n)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    dev = eng.upload(bundle)
    for keep in modes:
        best = None
        for rep in range(3):
            res = eng.trace_device(dev, 100, record_segments=keep)
            k, tot, nl = eng.result_timing(res)
            size = eng.result_size(res)
            eng.free_result(res)
            best = tot if best is None or tot < best else best
        print("rays %9d  log %-5s  solve %9.3f ms  launches %d  segments %d  hits %d  -> %.3e intersections/s" % (n, keep, best, nl, size[1], size[3], size[0] / (best * 1e-3)),
              flush=True)
    eng.free_batch(dev); eng.close()
