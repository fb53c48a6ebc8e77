"""The reference's Mesh kinematics KAT (test/runtests.jl:698-817): vertex transforms of translate / rotate / reset / align / scale
(Mesh.jl:78-167).  Host builders: the engine's triangle table is these vertices."""
import math

import numpy as np

import bmo_amd as bmo
from bmo_amd import shapes as sh

R2 = math.sqrt(2) / 2


def approx(a, b):
    return np.allclose(a, b, rtol=1.4901161193847656e-8, atol=0)


def extents(m):
    return m.vertices.min(axis=0), m.vertices.max(axis=0)


def test_mesh_kinematics():
    foo, bar = sh.CubeMesh(1), sh.CubeMesh(1)
    to_origin = -0.5 * np.ones(3)
    sh.translate3d(foo, to_origin)  # :718-727
    lo, hi = extents(foo)
    assert np.array_equal(lo, [-0.5] * 3) and np.array_equal(hi, [0.5] * 3) and np.all(foo.pos == -0.5)
    foo.set_new_origin3d()  # :729-732
    assert np.array_equal(foo.pos, np.zeros(3))
    sh.rotate3d(foo, [1, 0, 0], math.pi / 4)  # :735-745
    lo, hi = extents(foo)
    assert approx(lo, [-0.5, -R2, -R2]) and approx(hi, [0.5, R2, R2])
    sh.rotate3d(foo, [1, 0, 0], -math.pi / 4)
    sh.xrotate3d(foo, math.pi / 4)  # :747-755
    lo, hi = extents(foo)
    assert approx(lo, [-0.5, -R2, -R2]) and approx(hi, [0.5, R2, R2])
    sh.yrotate3d(foo, math.pi / 2)  # :757-765
    lo, hi = extents(foo)
    assert approx(lo, [-R2, -R2, -0.5]) and approx(hi, [R2, R2, 0.5])
    sh.zrotate3d(foo, math.pi / 4)  # :767-775
    lo, hi = extents(foo)
    assert approx(lo, [-0.5] * 3) and approx(hi, [0.5] * 3)
    d = foo.dir.T.reshape(-1)  # Julia's column-major linear indices 3, 5, 7 (:778)
    assert np.array_equal(d[[2, 4, 6]], [-1, 1, 1])
    sh.translate3d(bar, to_origin)
    bar.set_new_origin3d()
    sh.translate3d(foo, [1, 2, 3])  # :785-792
    sh.reset_translation3d(foo)
    sh.reset_rotation3d(foo)
    assert np.array_equal(foo.pos, np.zeros(3)) and approx(foo.dir, bar.dir)
    assert np.allclose(foo.vertices, bar.vertices, rtol=1.5e-8, atol=1e-15)
    sh.align3d(foo, np.array([0, 1, 1]) / math.sqrt(2))  # :794-801
    assert np.array_equal(foo.pos, np.zeros(3))
    assert np.allclose(foo.dir[:, 0], [1, 0, 0], atol=1e-15) and np.allclose(foo.dir[:, 1], [0, R2, R2], atol=1e-15)
    assert np.allclose(foo.dir[:, 2], [0, -R2, R2], atol=1e-15)
    sh.reset_rotation3d(foo)
    assert np.allclose(foo.normal3d(0), [0, 0, -1], atol=1e-15)  # :803-806
    foo.scale3d(2)  # :808-817
    lo, hi = extents(foo)
    assert approx(lo, [-1] * 3) and approx(hi, [1] * 3) and foo.scale == 2
