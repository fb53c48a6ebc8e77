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


def test_inexact_sdfs_are_bounded_through_their_slope_bound(oracle):
    """Aspheric / acylindric "sdfs" are first-order estimates, not distances.  The cull may still skip them because
    sdf(p) >= dist(p, solid) / K with K = sqrt(1 + max |z'|^2) (shapes.py slope_bound, derived there), and the bounding sphere of
    such a shape — and of every union / meniscus that contains one — is inflated by K x the usual margin.  Checked here on random
    lenses from the fuzz menus: K covers the sampled slopes, and outside the sphere the estimate never falls below dist / K."""
    from test_fuzz import _make

    rng_np = np.random.Generator(np.random.PCG64(2025))
    rng = np.random.default_rng(5)
    seen = 0
    for kind in ["asphere"] * 6 + ["asphere2"] * 6 + ["acylinder"] * 6:
        lens, _ = _make(rng_np, kind)
        bmo.xrotate3d(lens, 0.2)
        bmo.translate3d(lens, [1 * mm, -3 * mm, 2 * mm])
        sc = bmo.CompiledScene(bmo.System([lens]), [1e-6])
        top = lens.shape
        for s in [top] + list(getattr(top, "sdfs", [])):
            rec = sc._shapes[sc.shape_id(s)]
            if not (rec.flags & 1):
                continue
            c0, r0 = s.world_bound()
            kids = getattr(s, "sdfs", [s])
            ks = [k.slope_bound() if hasattr(k, "slope_bound") else 1.0 for k in kids]
            if any(k is None for k in ks):
                assert rec.bs_radius < 0  # no bound => never culled
                continue
            K = max(ks)
            assert K >= 1.0 and rec.bs_radius >= r0 + K * 1e-6
            for k in kids:  # K covers the profile's slope wherever it is sampled
                if hasattr(k, "slope_bound"):
                    for r in rng.uniform(0, k.diameter / 2, 64):
                        g = bmo.shapes.gradient_aspheric_equation(float(r), 1 / k.radius, k.conic_constant, k.coefficients)
                        assert math.sqrt(1 + g * g) <= K
            c = np.array(rec.bs_center[:])
            for _ in range(300):
                u = rng.normal(size=3)
                u /= np.linalg.norm(u)
                p = c + u * rec.bs_radius * (1 + 2 * rng.random())
                v = oracle.sdf(sc, s, p)
                assert v >= (np.linalg.norm(p - np.asarray(c0)) - r0) / K * (1 - 1e-9), (kind, type(s).__name__)
                assert v >= 1e-6  # what the cull needs: far above the reference's 1e-9 / 1e-10 thresholds
            seen += 1
    assert seen >= 18
