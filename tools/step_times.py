"""Per-launch kernel times of one workload of bench.py (BMO_DEBUG prints them): python tools/step_times.py c2v [rays]"""
import os, sys
os.environ["BMO_DEBUG"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_RAYS[name]
case = bench.Case(bmo, name, n, 0)
for rep in range(2):
    print("---- solve", rep, file=sys.stderr, flush=True)
    res, kms, nl = case.solve(100)
    print("kernel ms", kms, "launches", nl, case.eng.result_size(res), file=sys.stderr, flush=True)
    case.eng.free_result(res)
case.close()
