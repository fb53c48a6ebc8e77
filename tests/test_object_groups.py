"""The reference's "Object groups" KAT (test/runtests.jl:1095-1184): nested ObjectGroup kinematics — translate_to3d!, rotate3d!
about the group centre, reset_translation3d! / reset_rotation3d!, and Leaves iteration by System.  Host scene builders:
the GPU engine only ever sees the poses these functions leave behind, so their order of operations is parity-relevant."""
import math

import numpy as np

import bmo_amd as bmo


def point_object(position):
    """GroupTestObject(TestPoint(position)): an object whose shape is just a pose (runtests.jl:1096-1110)."""
    shape = bmo.shapes.AbstractShape()
    shape.pos = np.array(position, dtype=np.float64)
    return bmo.IntersectableObject(shape)


def test_object_group_kinematics():
    n = 8
    ang = bmo.linalg.linrange(0, 2 * math.pi * (n - 1) / n, n)
    xs, ys = np.cos(ang), np.sin(ang)
    center = point_object(np.zeros(3))
    circle = bmo.ObjectGroup([point_object([xs[i], ys[i], 0]) for i in range(n)])
    objects = bmo.ObjectGroup([center, circle])

    target = np.array([3.0, 0, 0])
    bmo.translate_to3d(objects, target)  # :1123-1136
    assert np.array_equal(objects.position(), target) and np.array_equal(center.position(), target) and np.array_equal(circle.position(), target)
    for i, obj in enumerate(circle.objects):
        assert np.array_equal(obj.position(), np.array([xs[i], ys[i], 0]) + target)

    angle = 2 * math.pi / n
    bmo.rotate3d(objects, [0, 0, 1], angle)  # :1138-1155
    Rt = bmo.linalg.rotate3d([0, 0, 1], angle)
    xt, yt = np.roll(xs, -1), np.roll(ys, -1)
    assert np.array_equal(objects.orientation(), Rt) and np.array_equal(circle.orientation(), Rt)
    assert np.allclose(center.orientation(), Rt, rtol=1.5e-8, atol=0)
    for i, obj in enumerate(circle.objects):
        assert np.array_equal(obj.orientation(), Rt)
        assert np.allclose(obj.position(), np.array([xt[i], yt[i], 0]) + target, rtol=1.5e-8, atol=1e-15)

    bmo.reset_translation3d(objects)  # :1157-1171
    bmo.reset_rotation3d(objects)
    Ri = np.eye(3)
    assert np.array_equal(objects.position(), np.zeros(3)) and np.array_equal(center.position(), np.zeros(3)) and np.array_equal(circle.position(), np.zeros(3))
    assert np.array_equal(objects.orientation(), Ri)
    assert np.allclose(center.orientation(), Ri, atol=1e-15) and np.allclose(circle.orientation(), Ri, atol=1e-15)
    for i, obj in enumerate(bmo.components.leaves(circle.objects)):
        assert abs(obj.position()[0] - xs[i]) <= 5e-16 and abs(obj.position()[1] - ys[i]) <= 5e-16

    system = bmo.System(objects)  # :1173-1183: only the leaves are exposed
    objs = system.objects()
    assert len(objs) == n + 1 and all(isinstance(o, bmo.IntersectableObject) for o in objs)
