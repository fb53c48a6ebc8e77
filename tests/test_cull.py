"""The engine's only algorithmic shortcut — the miss cull / nearest-hit prune — must be result-preserving.

CPU part (no GPU): (1) every SDF kind is bounded below by the distance to its bounding sphere (the premise of the
cull), checked on random points; (2) the lane code with cull tables == the lane code without == the oracle (which
never culls and burns the reference's 1000 iterations) on grazing / missing / inside-start rays.
"""
import math

import numpy as np
import pytest

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import c1_scene, c2_scene, disc_bundle, mm


def _shapes():
    out = [bmo.SphereSDF(3 * mm), bmo.PlanoSurfaceSDF(2 * mm, 8 * mm), bmo.ConvexSphericalSurfaceSDF(10 * mm, 8 * mm),
           bmo.ConcaveSphericalSurfaceSDF(10 * mm, 8 * mm), bmo.BoxSDF(3 * mm, 4 * mm, 5 * mm), bmo.CylinderSDF(2 * mm, 3 * mm),
           bmo.CutSphereSDF(5 * mm, 2 * mm), bmo.RingSDF(4 * mm, 1 * mm, 2 * mm), bmo.RightAnglePrismSDF(6 * mm, 4 * mm),
           bmo.BiConvexLensSDF(20 * mm, 30 * mm, 4 * mm, 10 * mm), bmo.PlanoConcaveLensSDF(15 * mm, 3 * mm, 10 * mm, 12 * mm),
           bmo.lens_shape_from_surfaces(bmo.SphericalSurface(20 * mm, 10 * mm), bmo.SphericalSurface(12 * mm, 10 * mm), 2 * mm)]
    for i, s in enumerate(out):
        bmo.translate3d(s, [1 * mm * i, -2 * mm, 3 * mm])
        bmo.xrotate3d(s, 0.3 + 0.1 * i)
        bmo.zrotate3d(s, -0.7)
    return out


def test_sdf_bounded_by_bounding_sphere(oracle):
    rng = np.random.default_rng(7)
    for s in _shapes():
        sc = bmo.CompiledScene(bmo.System([bmo.IntersectableObject(s)]), [1e-6])
        rec = sc._shapes[sc.shape_id(s)]
        c, R = np.array(rec.bs_center[:]), rec.bs_radius
        assert R > 0
        for _ in range(400):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            p = c + d * R * (1 + 3 * rng.random())
            assert oracle.sdf(sc, s, p) >= np.linalg.norm(p - c) - R, type(s).__name__


def _stress_bundle(n, seed):
    """Rays from all around a scene: many miss everything, some graze, some start inside lenses."""
    rng = np.random.default_rng(seed)
    pos = rng.normal(size=(n, 3)) * np.array([6 * mm, 25 * mm, 6 * mm]) + np.array([0, 20 * mm, 0])
    tgt = rng.normal(size=(n, 3)) * np.array([8 * mm, 30 * mm, 8 * mm]) + np.array([0, 20 * mm, 0])
    return bmo.RayBundle.rays(pos, tgt - pos, 1.064e-6)


@pytest.mark.parametrize("scene_fn", [c1_scene, c2_scene])
def test_cull_is_result_preserving(oracle, scene_fn):
    system, _ = scene_fn()
    b = _stress_bundle(300, 11)
    on = bmo.CompiledScene(system, b.lambdas, cull=True)
    off = bmo.CompiledScene(system, b.lambdas, cull=False)
    ref = oracle.trace(off, b, 30, threads=8)  # the oracle ignores bounding spheres: pure reference algorithm
    a = emu_trace(on, b, 30)
    c = emu_trace(off, b, 30)
    compare(a, ref, 0.0, "cull-on vs oracle")
    compare(c, ref, 0.0, "cull-off vs oracle")
    assert (ref.node_status & 1).sum() > 50  # plenty of genuine misses in the sample


def test_grazing_rays(oracle):
    """Rays tangent to a lens rim within +-1e-7 m: hit/miss decisions must match the 1000-iteration reference."""
    lens = bmo.SphericalLens(30 * mm, -30 * mm, 5 * mm, 12 * mm, 1.5)
    system = bmo.System([lens, bmo.Spotdetector(40 * mm)])
    bmo.translate3d(system.objects()[1], [0, 40 * mm, 0])
    offs = 6 * mm + np.linspace(-1e-7, 1e-7, 41)
    pos = np.stack([offs, -10 * mm * np.ones_like(offs), np.zeros_like(offs)], axis=1)
    b = bmo.RayBundle.rays(pos, np.tile([0.0, 1.0, 0.0], (len(offs), 1)), 1e-6)
    sc = bmo.CompiledScene(system, b.lambdas)
    compare(emu_trace(sc, b, 20), oracle.trace(sc, b, 20), 0.0, "grazing")
