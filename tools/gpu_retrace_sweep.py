"""Wide randomised retrace sweep on the GPU box: random scene, solve, random small move of one element, retrace; engine vs oracle.
usage: gpu_retrace_sweep.py [n_cases] [rays] [first seed]"""
import os, resource, signal, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_fuzz as f
import bmo_amd as bmo
import pyoracle
from parity import compare

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
base = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
pyoracle.lib()
cases = [(base + i, ("ray", "ray", "gauss", "pol")[i % 4]) for i in range(n_cases)]
t0 = time.time()
bad, stale, runaway, big, done = [], 0, 0, 0, 0
for i, (seed, kind) in enumerate(cases):
    n = n_rays if kind == "ray" else n_rays // 4
    try:
        scene0, scene1, bundle = f._retrace_case(seed, kind, n)
    except ValueError:
        continue
    try:
        g0, h0 = bmo.system._engine_solve(scene0, bundle, f.R_MAX, None, max_beams=300 * n)
        g1, h1 = bmo.system._engine_solve(scene1, bundle, f.R_MAX, h0, max_beams=300 * n)
    except RuntimeError as e:
        if "(-6)" not in str(e):
            raise
        runaway += 1
        continue
    if max(g0.n_intersect_calls, g1.n_intersect_calls) > 3_000_000:
        big += 1
        h0.free(); h1.free()
        continue
    a0, sol = pyoracle.trace(scene0, bundle, f.R_MAX, threads=16, keep=True)
    a1 = pyoracle.trace(scene1, bundle, f.R_MAX, threads=16, prev=sol)
    sol.free()
    if (a1.node_status & 512).any():
        stale += 1  # (round 4: compared like every other draw — kept stale children and stale-tail splits are the reference's result)
    try:
        compare(g0, a0, f._tol(kind), "first %d %s" % (seed, kind))
        compare(g1, a1, f._tol(kind), "retrace %d %s" % (seed, kind))
        done += 1
    except AssertionError as e:
        bad.append((seed, kind))
        print("FAIL", seed, kind, str(e)[:400], flush=True)
    h0.free(); h1.free()
    if i % 20 == 19:
        print("  %d / %d cases, %.0f s, %d failures" % (i + 1, len(cases), time.time() - t0, len(bad)), flush=True)
print("done: %d retraces compared (%d of them with BMO_NODE_RETRACE_STALE situations, %d runaway, %d too large for the oracle), failures: %s" % (done, stale, runaway, big, bad), flush=True)
sys.exit(1 if bad else 0)
