"""Runs tools/gpu_debug.py-style single traces in subprocesses with timeouts under different env settings."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, time
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bmo_amd as bmo
from scenes import c1_scene, c1_bundle
import pyoracle
from parity import compare
n = int(sys.argv[1])
system, _ = c1_scene(); bundle = c1_bundle(n)
scene = bmo.CompiledScene(system, bundle.lambdas)
eng = bmo.Engine(scene, 0)
print("tracing", n, flush=True)
got = eng.trace(bundle, 100)
print("traced kernel_ms", got.kernel_ms, "steps", got.n_steps, flush=True)
ref = pyoracle.trace(scene, bundle, 100, threads=8)
try:
    compare(got, ref, 0.0, "c1"); print("BIT-EXACT", flush=True)
except AssertionError as e:
    print("MISMATCH", str(e)[:500], flush=True)
''' % ROOT
log = open(os.path.join(ROOT, "gpurun_out", "debug2.log"), "w")
for env_extra, n in (({"BMO_DEBUG": "1"}, 64), ({}, 1000)):
    env = dict(os.environ); env.update(env_extra)
    log.write("=== %s n=%d\n" % (env_extra, n)); log.flush()
    try:
        p = subprocess.run([sys.executable, "-c", child, str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=45)
        out = p.stdout.decode(errors="replace")
        log.write(out[-4000:] + "\nrc=%d\n" % p.returncode)
    except subprocess.TimeoutExpired as e:
        log.write((e.stdout or b"").decode(errors="replace")[-4000:] + "\nTIMEOUT\n")
    log.flush()
print(open(os.path.join(ROOT, "gpurun_out", "debug2.log")).read())
