"""Vector helpers of the reference's Utils layer, restated for the host side.

Follows src/Utils/LinearAlgebraUtils.jl (rotate3d :55-65, align3d :73-95, normal3d :25-29,
isparallel3d :6-8) and src/Utils/OpticUtils.jl (sag :153, check_sag :155-161).
Scene set-up only: nothing here is on the traced hot path.
"""
import math

import numpy as np

inch = 25.4e-3  # src/Constants.jl:8
Z_vacuum = 376.730313668  # src/Constants.jl:6


def vec3(v):
    a = np.asarray(v, dtype=np.float64).reshape(3).copy()
    return a


def normalize(v):
    """normalize(::Point3) = a ./ norm(a) (GeometryBasics 0.5)."""
    v = np.asarray(v, dtype=np.float64)
    n = math.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return v / n


def rotate3d(axis, theta):
    """Rodrigues rotation matrix, LinearAlgebraUtils.jl:55-65 (same element expressions)."""
    cost = math.cos(theta)
    sint = math.sin(theta)
    ux, uy, uz = (float(a) for a in axis)
    return np.array(
        [
            [cost + ux ** 2 * (1 - cost), ux * uy * (1 - cost) - uz * sint, ux * uz * (1 - cost) + uy * sint],
            [uy * ux * (1 - cost) + uz * sint, cost + uy ** 2 * (1 - cost), uy * uz * (1 - cost) - ux * sint],
            [uz * ux * (1 - cost) - uy * sint, uz * uy * (1 - cost) + ux * sint, cost + uz ** 2 * (1 - cost)],
        ],
        dtype=np.float64,
    )


def _isapprox(a, b, atol=0.0):
    rtol = 0.0 if atol > 0 else math.sqrt(np.finfo(np.float64).eps)
    return abs(a - b) <= max(atol, rtol * max(abs(a), abs(b)))


def align3d(start, target):
    """LinearAlgebraUtils.jl:73-95."""
    start = normalize(start)
    target = normalize(target)
    rx, ry, rz = np.cross(target, start)
    cosA = float(np.dot(start, target))
    if _isapprox(cosA, 1.0):
        return np.eye(3)
    if _isapprox(cosA, -1.0):
        return np.diag([-1.0, -1.0, 1.0])
    k = 1 / (1 + cosA)
    return np.array(
        [
            [rx ** 2 * k + cosA, rx * ry * k + rz, rx * rz * k - ry],
            [ry * rx * k - rz, ry ** 2 * k + cosA, ry * rz * k + rx],
            [rz * rx * k + ry, rz * ry * k - rx, rz ** 2 * k + cosA],
        ]
    )


def normal3d(target, reference):
    """LinearAlgebraUtils.jl:25-29: unit vector perpendicular to both (right-hand rule)."""
    return normalize(np.cross(vec3(target), vec3(reference)))


def perpendicular(v):
    """Deterministic stand-in for the reference's random normal3d(input) (:35-47)."""
    v = vec3(v)
    new = np.array([0.0, 1.0, 0.0]) if abs(v[0]) > 0.9 * np.linalg.norm(v) else np.array([1.0, 0.0, 0.0])
    new = new - np.dot(new, v) * v / np.linalg.norm(v) ** 2
    return normalize(new)


def sag(r, l):
    """OpticUtils.jl:153."""
    return r - math.sqrt(r ** 2 - 0.25 * l ** 2)


def check_sag(r, d):
    """OpticUtils.jl:155-161."""
    if abs(2 * r) < d:
        raise ValueError(f"Radius of curvature (r = {r}) must be >= than half the diameter (d = {d}) or an illegal shape results!")


def linrange(start, stop, n):
    """LinRange(start, stop, n) as Julia evaluates its elements: lerpi(j, d, a, b) = (1 - t)*a + t*b with t = j/d, d = max(n - 1, 1)
    (Base range.jl `lerpi`; part of Julia Base, which is not under /root/reference — see DESIGN.md "unpinned arithmetic")."""
    d = max(int(n) - 1, 1)
    t = np.arange(int(n), dtype=np.float64) / d
    return (1 - t) * float(start) + t * float(stop)
