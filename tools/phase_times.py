"""Host-side phase breakdown of one trace (BMO_DEBUG laps) for C2 / C3 at a given size."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import scenes
which, n = sys.argv[1], int(sys.argv[2])
system = scenes.c2_scene()[0]
bundle = scenes.c2_bundle(n) if which == "c2" else (scenes.c2_survey_bundle(n) if which == "c2s" else scenes.c3_bundle(n))
scene = bmo.CompiledScene(system, bundle.lambdas)
eng = bmo.Engine(scene, 0)
dev = eng.upload(bundle)
for rep in range(3):
    t = time.perf_counter()
    res = eng.trace_device(dev, 100)
    dt = time.perf_counter() - t
    k, tot, nl = eng.result_timing(res)
    print("%s n=%d wall %.3f ms kernels %.3f ms total(ev) %.3f ms launches %d" % (which, n, dt * 1e3, k, tot, nl), file=sys.stderr, flush=True)
    eng.free_result(res)
