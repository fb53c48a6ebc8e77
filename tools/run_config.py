"""One solve of a BASELINE config (for rocprofv3 passes): run_config.py c3 [log2 n]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes
which = sys.argv[1]
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
system, bundle = {"c2": (scenes.c2_scene()[0], scenes.c2_bundle), "c3": (scenes.c2_scene()[0], scenes.c3_bundle), "c4": (scenes.c4_scene()[0], scenes.c4_bundle),
                  "c5": (scenes.c5_scene()[0], scenes.c5_bundle)}[which]
bundle = bundle(n)
scene = bmo.CompiledScene(system, bundle.lambdas)
eng = bmo.Engine(scene, 0)
dev = eng.upload(bundle)
for rep in range(2):
    res = eng.trace_device(dev, 100)
    k, tot, nl = eng.result_timing(res)
    size = eng.result_size(res)
    eng.free_result(res)
print(which, "n", n, "kernels %.3f ms solve %.3f ms launches %d" % (k, tot, nl), "calls/segments/beams/hits", size)
