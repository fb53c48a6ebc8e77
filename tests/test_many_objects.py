"""Scenes with more than 64 leaf objects: trace_all walks the candidate table in chunks of 64 slots (one bit mask per chunk,
bmo_lane.hpp tracing_step), so a scene of this size crosses a chunk boundary.  The reference's 101-mirror multipass cell
(test/runtests.jl:1013-1062) with rays fanned over it, plus a lens, a plate beamsplitter (two candidates: coating, substrate) and a
detector placed BEHIND the 64th slot.  Lane code (CPU) and engine (GPU) against the oracle, bit-exact."""
import math

import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import mm


def _cell():
    n_mirrors, radius = 101, 1.0
    L = 6 * radius / n_mirrors
    dth = 360 / (n_mirrors + 1)
    mirrors = [bmo.SquarePlanoMirror2D(L) for _ in range(n_mirrors)]
    th = dth
    for m in mirrors:
        point = radius * np.array([math.cos(math.radians(th)), math.sin(math.radians(th)), 0])
        bmo.zrotate3d(m, math.radians(th))
        bmo.translate3d(m, point)
        th += dth
    for m in mirrors:
        bmo.zrotate3d(m, math.radians(90))
    d = bmo.rotation_matrix([0, 0, 1], math.radians(dth)) @ np.array([-1.0, 0, 0])
    origin = np.array([radius, 0, 0]) - d
    # behind the mirrors in Leaves order (slots >= 101): a lens in the first chord of the cell, a plate splitter and a detector
    lens = bmo.SphericalLens(0.8, -0.8, 20 * mm, 50 * mm, 1.5)
    bmo.zrotate3d(lens, math.atan2(d[1], d[0]) - math.pi / 2)
    bmo.translate3d(lens, origin + 0.5 * d)
    plate = bmo.RectangularPlateBeamsplitter(60 * mm, 60 * mm, 5 * mm, lambda lam: 1.45)
    bmo.zrotate3d(plate, math.atan2(d[1], d[0]) - math.pi / 2 + math.radians(20))
    bmo.translate3d(plate, origin + 0.75 * d)
    det = bmo.Spotdetector(0.5)
    bmo.zrotate3d(det, math.atan2(d[1], d[0]) + math.pi / 2 + math.radians(40))
    bmo.translate3d(det, origin + 0.75 * d + 0.2 * np.array([-d[1], d[0], 0]))
    return bmo.System(mirrors + [lens, plate, det]), origin, d


def _bundle(n, origin, d):
    rng = np.random.Generator(np.random.PCG64(77))
    e1 = np.array([-d[1], d[0], 0.0])
    pos = origin[None, :] + (rng.uniform(-8, 8, n) * mm)[:, None] * e1[None, :] + (rng.uniform(-8, 8, n) * mm)[:, None] * np.array([0, 0, 1.0])[None, :]
    dirs = d[None, :] + rng.normal(scale=2e-3, size=(n, 3))
    return bmo.RayBundle.rays(pos, dirs, 1.064e-6)


def test_many_objects_lane_code(oracle):
    system, origin, d = _cell()
    b = _bundle(96, origin, d)
    scene = bmo.CompiledScene(system, b.lambdas)
    assert scene.n_objects == 104
    ref = oracle.trace(scene, b, 200, threads=8)
    assert ref.n_nodes > b.n and ref.node_nseg.max() > 20  # the plate splits, the cell folds the beams many times
    compare(emu_trace(scene, b, 200), ref, 0.0, "104 objects, lane code")


@pytest.mark.gpu
def test_many_objects_engine(oracle):
    assert bmo.abi.load_engine().bmo_device_count() >= 1, "no HIP device visible"
    system, origin, d = _cell()
    b = _bundle(4096, origin, d)
    scene = bmo.CompiledScene(system, b.lambdas)
    ref = oracle.trace(scene, b, 200, threads=16)
    eng = bmo.Engine(scene, 0)
    try:
        compare(eng.trace(b, 200), ref, 0.0, "104 objects, engine")
    finally:
        eng.close()
