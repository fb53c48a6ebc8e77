"""Time bmo_psf_intensity: n x n grid x H synthetic hits (pairs/s = n^2 * H / kernel time)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import bmo_amd as bmo

rng = np.random.Generator(np.random.PCG64(7))
for n, H in [(500, 1000), (100, 1 << 16), (100, 1 << 20), (1000, 1 << 14)]:
    hits = np.zeros((H, 9))
    hits[:, 0] = 1e-4 * rng.standard_normal(H); hits[:, 2] = 1e-4 * rng.standard_normal(H); hits[:, 1] = 0.2
    d = np.stack([1e-2 * rng.standard_normal(H), np.ones(H), 1e-2 * rng.standard_normal(H)], axis=1)
    hits[:, 3:6] = d / np.linalg.norm(d, axis=1)[:, None]
    hits[:, 6] = 0.21 + 1e-6 * rng.random(H); hits[:, 7] = np.abs(hits[:, 4]); hits[:, 8] = 2 * np.pi / 1e-6
    xs = bmo.linalg.linrange(-5e-5, 5e-5, n)
    best = 1e9
    for rep in range(3):
        I, _, ms = bmo.abi.psf_intensity(hits, [0, 0.2, 0], [1, 0, 0], [0, 0, 1], xs, xs)
        best = min(best, ms)
    print("psf n=%4d hits=%8d  kernel %9.3f ms  %.3e pairs/s" % (n, H, best, n * n * H / (best * 1e-3)), flush=True)
