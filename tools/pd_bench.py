"""Time bmo_photodetector_field on the C3-like scene (tests/test_photodetector.pd_scene): pairs/s = grid points x beamlets / kernel time."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bmo_amd as bmo
from test_photodetector import pd_scene

for n, res in [(4096, 100), (65536, 100), (65536, 256)]:
    system, pd, bundle = pd_scene(n)
    pd.resolution(res)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    g, gsol = bmo.system._engine_solve(scene, bundle, 20, None)
    hits = int(g.det_count[0]) // 3
    best = 1e9
    for rep in range(3):
        f = np.zeros((res, res), dtype=np.complex128)
        ms = gsol.photodetector_field(0, pd.position(), pd.orientation(), pd.x, pd.y, f)
        best = min(best, ms)
    print("pd grid %4d^2 beamlets %7d  all kernels %9.3f ms  %.3e pairs/s  |field|max %.3e" % (res, hits, best, res * res * hits / (best * 1e-3), np.abs(f).max()), flush=True)
    gsol.free()
