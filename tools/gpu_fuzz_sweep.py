"""Wide randomised parity sweep on the GPU box (beyond the seeds pinned in tests/test_fuzz.py): random element trains x random
bundles at wave/block-filling sizes, HIP engine (C ABI) against the CPU oracle, bit-exact for geometric rays.

Phase 1 (before anything touches the GPU): every seed is tried with a small bundle on the host emulator in a forked child under a
time / memory limit; seeds whose beam tree explodes (a splitter facing a mirror multiplies beams without bound — the reference would
not terminate on them either) are dropped.   Phase 2: engine vs oracle on the remaining seeds.
usage: gpu_fuzz_sweep.py [n_ray_seeds] [rays_per_bundle] [v|q] [seed offset]
"""
import os, resource, signal, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_fuzz as f
import bmo_amd as bmo
import pyoracle
from parity import emu_trace, compare

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
verbose = len(sys.argv) > 3 and sys.argv[3] == "v"
base = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # seed offset: another family of scenes
cases = [(s, "ray", n_rays) for s in range(base + 20000, base + 20000 + n_seeds)] + [(s, "pol", n_rays // 4) for s in range(base + 30000, base + 30000 + n_seeds // 4)] + \
        [(s, "gauss", n_rays // 4) for s in range(base + 40000, base + 40000 + n_seeds // 4)]
pyoracle.lib()
sc, bu = f._case(101, "ray", 8)
emu_trace(sc, bu, 5)  # load the emulator before forking

safe = []
for seed, kind, n in cases:
    pid = os.fork()
    if pid == 0:
        resource.setrlimit(resource.RLIMIT_AS, (4 << 30, 4 << 30))
        signal.alarm(10)
        try:
            scene, bundle = f._case(seed, kind, 64)
            got = emu_trace(scene, bundle, f.R_MAX)
            os._exit(0 if got.n_intersect_calls <= 64 * 400 else 2)
        except BaseException:
            os._exit(3)
    _, status = os.waitpid(pid, 0)
    if os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0:
        safe.append((seed, kind, n))
print("phase 1: %d of %d seeds kept" % (len(safe), len(cases)), flush=True)

t0 = time.time()
bad = []
calls = 0
runaway = 0
too_big = 0
skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0
stop = int(sys.argv[6]) if len(sys.argv) > 6 else len(safe)
for i, (seed, kind, n) in enumerate(safe):
    if i < skip or i >= stop:
        continue
    scene, bundle = f._case(seed, kind, n)
    if verbose:
        print("  case %d seed %d %s: start" % (i, seed, kind), flush=True)
    t1 = time.time()
    eng = bmo.Engine(scene, 0, max_beams=600 * n)
    try:
        got = eng.trace(bundle, f.R_MAX)
    except RuntimeError as e:
        if "(-6)" not in str(e):
            raise
        runaway += 1  # beams multiply without bound on this geometry (BMO_ERR_LIMIT): not a case the oracle can finish either
        continue
    finally:
        eng.close()
    t2 = time.time()
    if verbose:
        print("  seed %d %s: engine %.2f s, %d beams, %d segments, %d calls" % (seed, kind, t2 - t1, got.n_nodes, got.n_records, got.n_intersect_calls), flush=True)
    if got.n_intersect_calls > 4_000_000:  # a near-runaway tree under one root: minutes on the (single-threaded per root) oracle
        too_big += 1
        continue
    ref = pyoracle.trace(scene, bundle, f.R_MAX, threads=16)
    if verbose:
        print("      oracle %.2f s" % (time.time() - t2), flush=True)
    calls += ref.n_intersect_calls
    try:
        compare(got, ref, f._tol(kind), "sweep %d %s" % (seed, kind))
    except AssertionError as e:
        bad.append((seed, kind))
        print("FAIL", seed, kind, str(e)[:400], flush=True)
    if i % 20 == 19:
        print("  %d / %d cases, %.0f s, %d failures" % (i + 1, len(safe), time.time() - t0, len(bad)), flush=True)
print("done: %d cases compared (%d more stopped by max_beams, %d too large for the oracle), %.3e reference intersect3d calls compared, failures: %s" %
      (min(stop, len(safe)) - skip - runaway - too_big, runaway, too_big, calls, bad),
      flush=True)
sys.exit(1 if bad else 0)
