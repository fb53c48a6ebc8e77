"""Bundles mixing wavelengths through dispersive glass: every beam carries its own wavelength index into the per-object
refractive-index tables (the reference evaluates `n(λ)` per ray, Lenses.jl / RefractiveIndex).  Lane code == oracle, engine == oracle."""
import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import c2_bundle, c2_scene, c3_bundle


def mixed_bundle(kind, n):
    mk = c2_bundle if kind == "ray" else c3_bundle
    a, b = mk(n), mk(n)
    lam_plane = 6 if kind == "ray" else 18
    pb = b.planes.copy()
    pb[lam_plane] = 532e-9  # NBK7 table of the scene: 1.5195 at 532 nm, 1.5066 at 1064 nm (ucla.jl:5)
    planes = np.empty((a.planes.shape[0], 2 * n))
    planes[:, 0::2] = a.planes  # interleaved, so that neighbouring lanes of a wave differ in wavelength
    planes[:, 1::2] = pb
    return bmo.RayBundle(a.kind, planes)


@pytest.mark.parametrize("kind", ["ray", "gauss"])
def test_lane_code_mixed_wavelengths(oracle, kind):
    system, _ = c2_scene()
    bundle = mixed_bundle(kind, 48)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    assert scene.desc.n_lambda == 2
    ref = oracle.trace(scene, bundle, 100, threads=4)
    compare(emu_trace(scene, bundle, 100), ref, 0.0, "mixed lambda " + kind)
    # dispersion is really in play: the two colours of one start ray end at different detector positions
    h = ref.detector_hits(0)
    assert len(h) >= 2 and not np.allclose(h[0, :3], h[1, :3], rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ray", "gauss"])
def test_engine_mixed_wavelengths(oracle, kind):
    system, _ = c2_scene()
    bundle = mixed_bundle(kind, 2048 if kind == "ray" else 512)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    ref = oracle.trace(scene, bundle, 100, threads=16)
    eng = bmo.Engine(scene, 0)
    try:
        got = eng.trace(bundle, 100)
    finally:
        eng.close()
    compare(got, ref, 0.0, "mixed lambda gpu " + kind)
