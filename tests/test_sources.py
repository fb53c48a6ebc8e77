"""The reference's beam-group KATs (test/runtests.jl:600-696): PointSource and CollimatedSource generate the batch axis of the
hot path (one Beam per ray).  Host builders; the basis vector the reference draws at random is fixed here."""
import math

import numpy as np
import pytest

import bmo_amd as bmo


def angle3d(a, b):
    return math.acos(min(1.0, max(-1.0, float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))))))


def strictly_increasing(x):
    return all(b > a for a, b in zip(x, x[1:]))


def test_point_source():  # :600-647
    lam, pos, d, alpha, num_rays, num_rings = 486.0e-9, np.array([0, -0.5, 0]), np.array([0.0, 1, 1]), math.radians(2), 1000, 10
    source = bmo.PointSource(pos, d, alpha, lam, num_rays=num_rays, num_rings=num_rings)
    assert source.NA == math.sin(alpha)  # numerical_aperture(alpha), OpticUtils
    first = source.beams[0].rays[0]
    assert np.array_equal(first.pos, pos) and np.allclose(first.dir, d / np.linalg.norm(d), rtol=1.5e-8)
    last = source.beams[-1].rays[0]
    assert abs(angle3d(d, last.dir) - alpha) <= 1e-14
    assert np.array_equal(last.pos, pos) and len(source.beams) == num_rays
    dirs = np.array([b.rays[0].dir for b in source.beams])
    angles = np.array([angle3d(d, x) for x in dirs])
    required = bmo.linalg.linrange(0, alpha, num_rings)
    generated = np.unique(np.round(angles, 11))
    assert len(generated) == num_rings and np.allclose(generated, required, rtol=1.5e-8, atol=1e-12)
    per_angle = [int(np.isclose(angles, r, rtol=1.5e-8, atol=1e-12).sum()) for r in required]
    assert per_angle[0] == 1 and strictly_increasing(per_angle)
    assert len(np.unique(dirs, axis=0)) == len(dirs)
    with pytest.raises(ValueError):
        bmo.PointSource(pos, d, 1.1 * math.pi, lam, num_rays=num_rays, num_rings=num_rings)
    with pytest.raises(ValueError):
        bmo.PointSource(pos, d, alpha, lam, num_rays=100, num_rings=10)


def test_collimated_source():  # :649-695
    pos, d, diameter, num_rays, num_rings = np.array([0, -0.5, 0]), np.array([0.0, 1, 0]), 2 * bmo.inch, 500, 5
    source = bmo.CollimatedSource(pos, d, diameter, num_rays=num_rays, num_rings=num_rings)
    first = source.beams[0].rays[0]
    assert np.array_equal(first.pos, pos) and np.array_equal(first.dir, d) and first.lam == 1e-6
    last = source.beams[-1].rays[0]
    assert np.array_equal(last.dir, d) and math.isclose(np.linalg.norm(last.pos - pos), diameter / 2, rel_tol=1.5e-8)
    assert len(source.beams) == num_rays
    positions = np.array([b.rays[0].pos for b in source.beams])
    radii = np.linalg.norm(positions - pos, axis=1)
    required = bmo.linalg.linrange(0, diameter / 2, num_rings)
    generated = np.unique(np.round(radii, 11))
    assert len(generated) == num_rings and np.allclose(generated, required, rtol=1.5e-8, atol=1e-12)
    per_radius = [int(np.isclose(radii, r, rtol=1.5e-8, atol=1e-12).sum()) for r in required]
    assert per_radius[0] == 1 and strictly_increasing(per_radius)
    assert len(np.unique(positions, axis=0)) == len(positions)
    with pytest.raises(ValueError):
        bmo.CollimatedSource(pos, d, diameter, num_rays=100, num_rings=10)
