"""Degenerate inputs: NaN / Inf / zero-length rays, rays starting on or inside a surface, exactly on-axis and grazing rays.  The
engine must neither hang nor diverge from the reference algorithm on them: lane code == oracle on CPU, engine == oracle on GPU."""
import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import c1_scene, c2_scene

mm = 1e-3


def special_bundle(kind_scene):
    nan, inf = float("nan"), float("inf")
    rows = [
        # pos, dir
        ([0, -0.05, 0], [0, 1, 0]),                 # exactly on axis (the reference's NaN-normal corner)
        ([nan, -0.05, 0], [0, 1, 0]),               # NaN position
        ([0, -0.05, 0], [nan, 1, 0]),               # NaN direction
        ([0, -0.05, 0], [0, 0, 0]),                 # zero direction (not normalisable)
        ([inf, -0.05, 0], [0, 1, 0]),               # infinite position
        ([0, -0.05, 0], [0, inf, 0]),               # infinite direction
        ([1e30, -0.05, 0], [-1, 0, 0]),             # astronomically far away
        ([0, 0.001, 0], [0, 1, 0]),                 # starts inside the first lens (C1) / near the object plane (C2)
        ([0, 0.0, 0], [0, 1, 0]),                   # starts on the vertex of the first surface
        ([0, -0.05, 12.7 * mm], [0, 1, 0]),         # along the rim of a 25.4 mm lens
        ([0, -0.05, 12.7 * mm - 1e-9], [0, 1, 0]),  # 1 nm inside the rim
        ([0, -0.05, 12.7 * mm + 1e-9], [0, 1, 0]),  # 1 nm outside the rim
        ([0.003, -0.05, 0.002], [0, -1, 0]),        # pointing away from everything
        ([0, -0.05, 0], [1e-300, 1, 0]),            # denormal-sized component
        ([0.001, -0.05, 0.001], [0, 1, 1e-17]),     # below-epsilon tilt
    ]
    pos = np.array([r[0] for r in rows], dtype=np.float64)
    d = np.array([r[1] for r in rows], dtype=np.float64)
    if kind_scene == "c2":  # C2's axis is +z, object plane at z = -0.77 mm
        pos = pos[:, [0, 2, 1]]
        d = d[:, [0, 2, 1]]
        pos[:, 2] = np.where(np.isfinite(pos[:, 2]), pos[:, 2] * 0.0154, pos[:, 2])  # -0.05 -> -0.77 mm
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        nrm = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        d = d / nrm[:, None]  # the Ray constructor normalises (Ray.jl); NaN / 0 / Inf inputs stay what the division makes them
    return bmo.RayBundle.rays(pos, d, 1.064e-6)


@pytest.mark.parametrize("which", ["c1", "c2"])
def test_lane_code_on_degenerate_rays(oracle, which):
    system = (c1_scene if which == "c1" else c2_scene)()[0]
    bundle = special_bundle(which)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    ref = oracle.trace(scene, bundle, 30, threads=2)
    assert ref.n_nodes >= bundle.n
    compare(emu_trace(scene, bundle, 30), ref, 0.0, "degenerate " + which)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["c1", "c2"])
def test_engine_on_degenerate_rays(oracle, which):
    system = (c1_scene if which == "c1" else c2_scene)()[0]
    bundle = special_bundle(which)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    ref = oracle.trace(scene, bundle, 30, threads=2)
    eng = bmo.Engine(scene, 0)
    try:
        got = eng.trace(bundle, 30)
    finally:
        eng.close()
    compare(got, ref, 0.0, "degenerate gpu " + which)
