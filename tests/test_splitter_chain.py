"""Several beam splitters met inside ONE fused launch: a splitting lane goes on with its transmitted child in place and pushes the reflected
one to the next launch while the launch's reservation lasts (StepParams::inwave_cap = one push per record of the launch); a chain of four
splitters asks every lane for four pushes, so the reservation runs out in mid-launch and the remaining splits take the end-of-loop path
(both children to the next launch).  Engine against the oracle for Ray, PolarizedRay and GaussianBeamlet bundles; beam tree of depth 4 (heap
ordering) and, with a fifth and sixth splitter, depth 6 (sort ordering)."""
import math

import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import mm


def _chain(n_split):
    objs = []
    lens = bmo.SphericalLens(120 * mm, -120 * mm, 4 * mm, 25.4 * mm, 1.5)
    bmo.translate3d(lens, [0, 10 * mm, 0])
    objs.append(lens)
    for q in range(n_split):
        bs = bmo.ThinBeamsplitter(30 * mm)
        bmo.xrotate3d(bs, math.radians(45 if q % 2 == 0 else -45))
        bmo.translate3d(bs, [0, (40 + 25 * q) * mm, 0])
        objs.append(bs)
    end = bmo.Spotdetector(40 * mm)
    bmo.translate3d(end, [0, (40 + 25 * n_split + 20) * mm, 0])
    objs.append(end)
    return bmo.System(objs)


def _bundle(kind, n):
    rng = np.random.Generator(np.random.PCG64(4242))
    pos = np.zeros((n, 3))
    pos[:, 0] = rng.uniform(-3, 3, n) * mm
    pos[:, 2] = rng.uniform(-3, 3, n) * mm
    dirs = np.tile(np.array([0.0, 1.0, 0.0]), (n, 1)) + rng.normal(scale=1e-3, size=(n, 3))
    if kind == "ray":
        return bmo.RayBundle.rays(pos, dirs, 1.064e-6)
    if kind == "pol":  # (axis-parallel: E0 stays exactly orthogonal to the direction)
        return bmo.RayBundle.from_beams([bmo.Beam(bmo.PolarizedRay(p, [0.0, 1.0, 0.0], 1.064e-6, [1.0, 0, 0])) for p in pos])
    beamlets = [bmo.GaussianBeamlet(p, d / np.linalg.norm(d), 1.064e-6, 0.3 * mm, support=[1, 0, 0]) for p, d in zip(pos, dirs)]
    return bmo.RayBundle.from_beams(beamlets)


@pytest.mark.parametrize("kind", ["ray", "gauss"])
def test_splitter_chain_lane_code(oracle, kind):
    system = _chain(4)
    b = _bundle(kind, 24)
    scene = bmo.CompiledScene(system, b.lambdas)
    ref = oracle.trace(scene, b, 50, threads=8)
    assert ref.n_nodes >= 24 * 9  # every level of the tree splits once more
    compare(emu_trace(scene, b, 50), ref, 0.0, f"splitter chain, lane code, {kind}")


@pytest.mark.gpu
@pytest.mark.parametrize("n_split", [4, 6])
@pytest.mark.parametrize("kind", ["ray", "pol", "gauss"])
def test_splitter_chain_engine(oracle, kind, n_split):
    assert bmo.abi.load_engine().bmo_device_count() >= 1, "no HIP device visible"
    system = _chain(n_split)
    b = _bundle(kind, 2048 if kind != "gauss" else 512)
    scene = bmo.CompiledScene(system, b.lambdas)
    ref = oracle.trace(scene, b, 50, threads=16)
    assert ref.n_nodes > 8 * b.n
    eng = bmo.Engine(scene, 0)
    try:
        compare(eng.trace(b, 50), ref, 0.0, f"splitter chain x{n_split}, engine, {kind}")
    finally:
        eng.close()
