import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bmo_amd as bmo, pyoracle as oracle
from test_retrace import retrace_pair
kind, case, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
s0, s1, b = retrace_pair(kind, case, n)
a0, sol = oracle.trace(s0, b, 20, keep=True)
a1 = oracle.trace(s1, b, 20, prev=sol)
g0, gs0 = bmo.system._engine_solve(s0, b, 20, None)
g1, gs1 = bmo.system._engine_solve(s1, b, 20, gs0)
print("calls oracle first/retrace", a0.n_intersect_calls, a1.n_intersect_calls, "gpu", g0.n_intersect_calls, g1.n_intersect_calls)
print("nodes", a1.n_nodes, g1.n_nodes, "recs", a1.n_records, g1.n_records, "steps", g0.n_steps, g1.n_steps)
print("status eq", np.array_equal(a1.node_status, g1.node_status), "obj eq", np.array_equal(a1.rec_obj, g1.rec_obj))
for rm in range(1, 12):
    a = oracle.trace(s1, b, rm, prev=sol)
    g, gs = bmo.system._engine_solve(s1, b, rm, gs0)
    print("r_max", rm, "calls oracle", a.n_intersect_calls, "gpu", g.n_intersect_calls, "nseg", a.node_nseg.tolist()[:6], g.node_nseg.tolist()[:6])
for rm in (1, 2):
    g, gs = bmo.system._engine_solve(s1, b, rm, gs0)
    print("r_max", rm, "status", g.node_status, "rec_obj", g.rec_obj, "shape", g.rec_shape, "t c/w/d", g.rec[7], g.rec[18], g.rec[29])
