"""Time the engine on the BASELINE scenes (C1..C5) for the library named by BMO_ENGINE_LIB (default: in-tree build)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
cases = [("c1", scenes.c1_scene()[0], scenes.c1_bundle(n), 20), ("c2", scenes.c2_scene()[0], scenes.c2_bundle(n), 20),
         ("c3", scenes.c2_scene()[0], scenes.c3_bundle(max(1, n // 32)), 20), ("c4", scenes.c4_scene()[0], scenes.c4_bundle(n), 20),
         ("c5", scenes.c5_scene()[0], scenes.c5_bundle(n // 4), 50)]
for name, system, bundle, r_max in cases:
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    dev = eng.upload(bundle)
    best = (1e9, 0, 0)
    for rep in range(3):
        res = eng.trace_device(dev, r_max)
        kms, tms, nl = eng.result_timing(res)
        eng.free_result(res)
        if kms < best[0]:
            best = (kms, tms, nl)
    eng.free_batch(dev)
    print("%-22s %-3s rays %8d  kernels %8.3f ms  total %8.3f ms  launches %3d" % (os.environ.get("BMO_ENGINE_LIB", "default")[-22:], name, bundle.n, *best), flush=True)
    eng.close()
