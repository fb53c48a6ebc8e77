"""First-contact GPU debug: progressive log so a hang can be located."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
LOG = open(os.path.join(ROOT, "gpurun_out", "debug.log"), "a")
def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush(); os.fsync(LOG.fileno())
log("start")
import numpy as np
import bmo_amd as bmo
log("imported")
lib = bmo.abi.load_engine()
log("engine loaded, devices:", lib.bmo_device_count())
from scenes import c1_scene, c1_bundle, c2_scene, c2_bundle
import pyoracle
from parity import compare
for name, mk, bun, n in (("c1", c1_scene, c1_bundle, 64), ("c1", c1_scene, c1_bundle, 1000), ("c2", c2_scene, c2_bundle, 256), ("c2", c2_scene, c2_bundle, 4096)):
    system, _ = mk()
    bundle = bun(n)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    log(name, n, "scene compiled; blob shapes", len(scene.shape_list))
    eng = bmo.Engine(scene, 0)
    log(" scene created")
    t = time.time()
    got = eng.trace(bundle, 100)
    log(" traced in %.3fs kernel_ms %.3f steps %d recs %d calls %d" % (time.time() - t, got.kernel_ms, got.n_steps, got.n_records, got.n_intersect_calls))
    eng.close()
    t = time.time()
    ref = pyoracle.trace(scene, bundle, 100, threads=16)
    log(" oracle in %.3fs" % (time.time() - t))
    try:
        compare(got, ref, 0.0, name)
        log(" BIT-EXACT")
    except AssertionError as e:
        log(" MISMATCH", str(e)[:600])
log("done")
