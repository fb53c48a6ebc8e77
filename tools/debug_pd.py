"""Find the beamlet whose Photodetector field differs between engine and oracle (seed given on the command line)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_fuzz as f
import bmo_amd as bmo
import pyoracle
pyoracle.lib()
seed = int(sys.argv[1])
scene, bundle = f._case(seed, "gauss", 512)
pd = scene.detectors[0]
print([type(o).__name__ for o in scene.leaf_objects])
def fields(b):
    a, osol = pyoracle.trace(scene, b, f.R_MAX, threads=16, keep=True)
    g, gsol = bmo.system._engine_solve(scene, b, f.R_MAX, None)
    fa = np.zeros((len(pd.x), len(pd.y)), dtype=np.complex128); fg = fa.copy()
    osol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fa)
    gsol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, fg)
    gsol.free(); osol.free()
    return a, g, fa, fg
a, g, fa, fg = fields(bundle)
print("all: hits", a.det_count, g.det_count, "max diff", np.abs(fa - fg).max(), "peak", np.abs(fa).max())
shown = 0
for i in range(bundle.n):
    b = bmo.RayBundle(bundle.kind, bundle.planes[:, i:i + 1].copy())
    a, g, fa, fg = fields(b)
    d = np.abs(fa - fg).max()
    if d > 1e-9 * max(np.abs(fa).max(), 1e-300) or (np.abs(fa).max() == 0) != (np.abs(fg).max() == 0):
        print("beamlet", i, "diff", d, "peak", np.abs(fa).max(), np.abs(fg).max(), "nodes", a.n_nodes, "hits", a.det_count, g.det_count)
        for j in range(a.n_nodes):
            fr, ns = a.node_first_rec[j], a.node_nseg[j]
            print("   node", j, "parent", a.node_parent[j], "status", a.node_status[j], "objs", a.rec_obj[fr:fr + ns].tolist(), "aux", a.node_aux[j].tolist())
        print("   det rows oracle", a.det_data[:6].tolist())
        print("   det rows engine", g.det_data[:6].tolist())
        shown += 1
        if shown >= 3:
            break
print("done")
