"""Host-side mirror of the reference's geometry layer (shapes + kinematics + shape builders).

This is scene SET-UP code: it builds the flat shape tables that the HIP
engine consumes.  No tracing happens here.  Each class/function cites the reference
definition it mirrors (paths relative to /root/reference/src).

Kinematics follow AbstractTypes/AbstractShape.jl:44-114, SDFs/AbstractSDF.jl:20-27,
SDFs/UnionSDF.jl:63-82 and Mesh.jl:78-175.
"""
import math

import numpy as np

from . import linalg as la

# shape kinds (include/bmo.h enum bmo_shape_kind)
(K_MESH, K_SPHERE, K_PLANO, K_CONVEX, K_CONCAVE, K_UNION, K_BOX, K_CYLINDER, K_CUTSPHERE, K_RING, K_PRISM, K_MENISCUS, K_POINT, K_ASPH_CONVEX,
 K_ASPH_CONCAVE, K_CYL_CONVEX, K_CYL_CONCAVE) = range(17)
FLAG_INEXACT = 1


class AbstractShape:
    """AbstractTypes/AbstractShape.jl:31"""

    def __init__(self):
        self.pos = np.zeros(3)
        self.dir = np.eye(3)

    # --- kinematics (AbstractShape.jl:44-114)
    def _set_orientation(self, d):
        self.dir = np.array(d, dtype=np.float64)

    def _translate3d(self, offset):
        self.pos = self.pos + la.vec3(offset)

    def _rotate3d(self, axis, theta):
        R = la.rotate3d(axis, theta)
        self._set_orientation(R @ self.dir)

    def _align3d(self, target_axis):
        R = la.align3d(self.dir[:, 1], target_axis)
        self._set_orientation(R @ self.dir)

    def _reset_translation3d(self):
        self.pos = np.zeros(3)

    def _reset_rotation3d(self):
        self._set_orientation(np.eye(3))


# ----------------------------------------------------------------------------- meshes
class Mesh(AbstractShape):
    """Mesh.jl:33-39.  `faces` is 0-based here (the reference is 1-based)."""

    kind = K_MESH

    def __init__(self, vertices, faces, scale=1.0):
        super().__init__()
        self.vertices = np.array(vertices, dtype=np.float64)
        self.faces = np.array(faces, dtype=np.int64)
        self.scale = float(scale)

    def _translate3d(self, offset):  # Mesh.jl:78-82
        off = la.vec3(offset)
        self.pos = self.pos + off
        self.vertices = self.vertices + off[None, :]

    def _rotate3d(self, axis, theta):  # Mesh.jl:89-96
        R = la.rotate3d(axis, theta)
        self.vertices = (self.vertices - self.pos[None, :]) @ R.T + self.pos[None, :]
        self.dir = R @ self.dir

    def _align3d(self, target_axis):  # Mesh.jl:114-121
        R = la.align3d(self.dir[:, 1], target_axis)
        self.vertices = (self.vertices - self.pos[None, :]) @ R.T + self.pos[None, :]
        self.dir = self.dir @ R

    def _reset_translation3d(self):  # Mesh.jl:139-142
        self._translate3d(-self.pos)

    def _reset_rotation3d(self):  # Mesh.jl:149-163: rotate the vertices back about the net rotation axis, then dir = I
        R = self.dir
        theta = math.acos(min(1.0, max(-1.0, (np.trace(R) - 1) / 2)))
        if theta == 0:
            return
        axis = 1 / (2 * math.sin(theta)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        self._rotate3d(axis, -theta)
        self.dir = np.eye(3)

    def scale3d(self, s):  # Mesh.jl:127-132
        self.vertices = (self.vertices - self.pos[None, :]) * s + self.pos[None, :]
        self.scale = float(s)

    def set_new_origin3d(self):  # Mesh.jl:170-174
        self.dir = np.eye(3)
        self.pos = np.zeros(3)

    def triangles(self):
        """(n_faces, 9) world-space V1 V2 V3 rows, as intersect3d reads them (Mesh.jl:252)."""
        return self.vertices[self.faces].reshape(len(self.faces), 9)

    def normal3d(self, fid):  # Mesh.jl:183-192
        f = self.vertices[self.faces[fid]]
        return la.normalize(np.cross(f[1] - f[0], f[2] - f[0]))


def RectangularFlatMesh(width, height):  # Mesh.jl:282-303
    x, z = width / 2, height / 2
    v = [[x, 0, z], [x, 0, -z], [-x, 0, -z], [-x, 0, z]]
    f = [[0, 1, 3], [1, 2, 3]]
    return Mesh(v, f)


def QuadraticFlatMesh(width):  # Mesh.jl:310
    return RectangularFlatMesh(width, width)


def CircularFlatMesh(radius, n=30):  # Mesh.jl:322-348
    ang = np.linspace(0, 2 * math.pi * (n - 1) / n, n)
    xs = np.concatenate([[0.0], np.cos(ang) * radius])
    zs = np.concatenate([[0.0], np.sin(ang) * radius])
    v = np.stack([xs, np.zeros(n + 1), zs], axis=1)
    f = np.array([[0, i, i + 1] for i in range(1, n + 1)], dtype=np.int64)
    f[-1, 2] = 1  # `faces[end] = 2` (1-based, column-major last element)
    return Mesh(v, f)


def CuboidMesh(x, y, z, theta=math.pi / 2):  # Mesh.jl:362-395
    dx = math.cos(theta) * y
    v = [[0, 0, 0], [x, 0, 0], [x + dx, y, 0], [0 + dx, y, 0], [0 + dx, y, z], [x + dx, y, z], [x, 0, z], [0, 0, z]]
    f1 = [[1, 3, 2], [1, 4, 3], [3, 4, 5], [3, 5, 6], [2, 3, 6], [2, 6, 7], [1, 8, 5], [1, 5, 4], [6, 5, 8], [6, 8, 7], [1, 7, 8], [1, 2, 7]]
    return Mesh(v, np.array(f1) - 1)


def CubeMesh(scale):  # Mesh.jl:398-401
    s = float(scale)
    return CuboidMesh(s, s, s)


def RetroMesh(scale=1.0):  # OpticalComponents/Misc.jl:8-22
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64) * scale
    f = np.array([[1, 3, 2], [1, 4, 3], [1, 2, 4]]) - 1
    return Mesh(v, f, scale)


# ------------------------------------------------------------------------------- SDFs
class AbstractSDF(AbstractShape):
    """SDFs/AbstractSDF.jl:18"""

    def __init__(self):
        super().__init__()
        self.tdir = np.eye(3)

    def _set_orientation(self, d):  # orientation! AbstractSDF.jl:20-23
        self.dir = np.array(d, dtype=np.float64)
        self.tdir = self.dir.T.copy()

    def __add__(self, other):  # UnionSDF.jl:58-61
        a = self.sdfs if isinstance(self, UnionSDF) else [self]
        b = other.sdfs if isinstance(other, UnionSDF) else [other]
        return UnionSDF(*(list(a) + list(b)))

    # local-frame bounding sphere (centre, radius) of the solid; used only for the miss cull
    def _local_bound(self):
        raise NotImplementedError

    def world_bound(self):
        c, r = self._local_bound()
        return self.pos + self.dir @ np.asarray(c, dtype=np.float64), float(r)

    def params(self):
        return []


class PlanoSurfaceSDF(AbstractSDF):  # SphericalLensSDF.jl:41-65
    kind = K_PLANO

    def __init__(self, thickness, diameter):
        super().__init__()
        self.thickness = float(thickness)
        self.diameter = float(diameter)

    def params(self):
        return [self.thickness, self.diameter]

    def _local_bound(self):
        t, d = self.thickness, self.diameter
        return [0, t / 2, 0], math.hypot(t / 2, d / 2)


class SphereSDF(AbstractSDF):  # SphericalLensSDF.jl:72-89 (orientation fixed to identity)
    kind = K_SPHERE

    def __init__(self, radius):
        super().__init__()
        self.radius = float(radius)

    def _set_orientation(self, d):  # orientation!(::SphereSDF, ::Any) = nothing
        pass

    @property
    def diameter(self):
        return 2 * self.radius

    @property
    def thickness(self):
        return 2 * self.radius

    def params(self):
        return [self.radius]

    def _local_bound(self):
        return [0, 0, 0], self.radius


class ConcaveSphericalSurfaceSDF(AbstractSDF):  # SphericalLensSDF.jl:131-170
    kind = K_CONCAVE

    def __init__(self, radius, diameter):
        super().__init__()
        la.check_sag(radius, diameter)
        self.radius = float(radius)
        self.diameter = float(diameter)
        self.sag = la.sag(radius, diameter)

    thickness = 0.0

    def params(self):
        return [self.radius, self.diameter, self.sag]

    def _local_bound(self):
        return [0, -self.sag / 2, 0], math.hypot(self.sag / 2, self.diameter / 2)


class ConvexSphericalSurfaceSDF(AbstractSDF):  # SphericalLensSDF.jl:186-232
    kind = K_CONVEX

    def __init__(self, radius, diameter):
        super().__init__()
        la.check_sag(radius, diameter)
        self.radius = float(radius)
        self.diameter = float(diameter)
        self.sag = la.sag(radius, diameter)
        self.height = self.radius - self.sag

    @property
    def thickness(self):
        return self.sag

    def params(self):
        return [self.radius, self.diameter, self.sag, self.height]

    def _local_bound(self):
        return [0, self.sag / 2, 0], math.hypot(self.sag / 2, self.diameter / 2)


class BoxSDF(AbstractSDF):  # PrimitiveSDF.jl:13-46
    kind = K_BOX

    def __init__(self, x, y, z):
        super().__init__()
        self.dimensions = np.array([x / 2, y / 2, z / 2], dtype=np.float64)

    @property
    def thickness(self):
        return 2 * self.dimensions[1]

    def params(self):
        return list(self.dimensions)

    def _local_bound(self):
        return [0, 0, 0], float(np.linalg.norm(self.dimensions))


class CylinderSDF(AbstractSDF):  # PrimitiveSDF.jl:53-76
    kind = K_CYLINDER

    def __init__(self, r, h):
        super().__init__()
        self.radius = float(r)
        self.height = float(h)

    def params(self):
        return [self.radius, self.height]

    def _local_bound(self):
        return [0, 0, 0], math.hypot(self.radius, self.height)


class CutSphereSDF(AbstractSDF):  # PrimitiveSDF.jl:83-124
    kind = K_CUTSPHERE

    def __init__(self, radius, height):
        super().__init__()
        if abs(height) >= radius:
            raise ValueError("Cut off height must be smaller than radius")
        self.radius = float(radius)
        self.height = float(height)
        self.w = math.sqrt(radius ** 2 - height ** 2)

    def params(self):
        return [self.radius, self.height, self.w]

    def _local_bound(self):
        return [0, 0, 0], self.radius


class RingSDF(AbstractSDF):  # PrimitiveSDF.jl:132-166
    kind = K_RING

    def __init__(self, inner_radius, width, thickness):
        super().__init__()
        self.inner_radius = inner_radius + width / 2
        self.hwidth = width / 2
        self.hthickness = thickness / 2

    def params(self):
        return [self.inner_radius, self.hwidth, self.hthickness]

    def _local_bound(self):
        return [0, 0, 0], math.hypot(self.inner_radius + self.hwidth, self.hthickness)


class RightAnglePrismSDF(AbstractSDF):  # PrimitiveSDF.jl:183-210
    kind = K_PRISM

    def __init__(self, leg_length, height):
        super().__init__()
        self.dimensions = np.array([leg_length / 2, leg_length / 2, height / 2], dtype=np.float64)

    def params(self):
        return list(self.dimensions)

    def _local_bound(self):
        return [0, 0, 0], float(np.linalg.norm(self.dimensions))


class TestPointSDF(AbstractSDF):  # test/runtests.jl:926-947 (KAT fixture)
    kind = K_POINT
    __test__ = False

    def _local_bound(self):
        return [0, 0, 0], 0.0


# ------------------------------------------------------------------- aspheres (AsphericalLensSDF.jl)
def _ipow(x, n):
    if n == 0:
        return 1.0
    if n <= 3:
        r = x
        for _ in range(n - 1):
            r = r * x
        return r
    r = 1.0
    while n > 0:
        if n & 1:
            r *= x
        x *= x
        n >>= 1
    return r


def aspheric_equation(r, c, k, coeffs):
    """AsphericalLensSDF.jl:133-141"""
    r2 = r * r
    sqrt_arg = 1 - (1 + k) * (c * c) * r2
    if sqrt_arg < 0:
        return math.nan
    sum_a = 0.0
    for i, a in enumerate(coeffs, start=1):
        term = a * _ipow(r2, i)
        sum_a = term if i == 1 else sum_a + term
    return c * r2 / (1 + math.sqrt(sqrt_arg)) + sum_a


def gradient_aspheric_equation(r, c, k, coeffs):
    """AsphericalLensSDF.jl:147-156 (first component; the second is 1)."""
    Ri = 1 / c
    sqrt_arg = 1 - (r * r) * (1 + k) / (Ri * Ri)
    if sqrt_arg < 0:
        return math.nan
    sq = math.sqrt(sqrt_arg)
    gr = 2 * r / (Ri * (sq + 1)) + (r * r * r) * (1 + k) / ((Ri * Ri * Ri) * sq * ((sq + 1) * (sq + 1)))
    sum_r = 0.0
    for m, a in enumerate(coeffs, start=1):
        term = 2 * m * a * _ipow(r, 2 * (m - 1) + 1)
        sum_r = term if m == 1 else sum_r + term
    return -sum_r - gr


def _find_zero_bisection(f, a, b, tol=1e-10, max_iter=1000):
    """Utils/MiscUtils.jl:87-108"""
    fa, fb = f(a), f(b)
    if _sign(fa) == _sign(fb):
        raise ValueError("Bisection requires a sign change")
    for _ in range(max_iter):
        mid = (a + b) / 2
        fmid = f(mid)
        if abs(fmid) < tol:
            return mid
        if _sign(fa) == _sign(fmid):
            a, fa = mid, fmid
        else:
            b, fb = mid, fmid
    raise ValueError("Bisection did not converge")


def max_aspheric_value(c, k, coeffs, d):
    """AsphericalLensSDF.jl:53-67"""
    f = lambda r: aspheric_equation(r, c, k, coeffs)
    fp = lambda r: gradient_aspheric_equation(r, c, k, coeffs)
    a, b = 1e-8, d / 2
    if _sign(fp(a)) == _sign(fp(b)):
        r_max = a if abs(f(a)) > abs(f(b)) else b
    else:
        r_max = _find_zero_bisection(fp, a, b)
    return f(r_max), r_max


def _profile_slope_bound(radius, conic_constant, coefficients, diameter, samples=4097):
    g_max = 0.0
    for r in np.linspace(0.0, diameter / 2, samples):
        g = gradient_aspheric_equation(float(r), 1 / radius, conic_constant, coefficients)
        z = aspheric_equation(float(r), 1 / radius, conic_constant, coefficients)
        if math.isnan(g) or math.isnan(z) or math.isinf(g):
            return None
        g_max = max(g_max, abs(g))
    g_max *= 1.05
    return math.sqrt(1 + g_max * g_max)


class _AsphericalSurfaceSDF(AbstractSDF):
    flags = FLAG_INEXACT  # first-order distance estimate: bounding-sphere culls only

    def __init__(self, coefficients, radius, conic_constant, diameter):
        super().__init__()
        self.coefficients = [float(x) for x in coefficients]
        self.radius = float(radius)
        self.conic_constant = float(conic_constant)
        self.diameter = float(diameter)
        self.max_sag = max_aspheric_value(1 / self.radius, self.conic_constant, self.coefficients, self.diameter)

    def edge_sag_value(self):
        return aspheric_equation(self.diameter / 2, 1 / self.radius, self.conic_constant, self.coefficients)

    def params(self):
        return [self.radius, self.conic_constant, self.diameter, self.max_sag[0]]

    def slope_bound(self):
        """K >= 1 with  sdf(p) >= dist(p, solid) / K  for every p outside the solid (None: no such bound, the shape is not culled).

        The aspheric "sdf" is a first-order estimate: |z - z(r)| / sqrt(1 + z'(r)^2) over the aperture, exact distances to the closing
        segments of the 2D perimeter divided by sqrt(1 + z'(d/2)^2) elsewhere, and the minimum of such terms
        (AsphericalLensSDF.jl:186-307).  |z - z(r)| is the distance to the surface point straight above / below p, so it is at
        least dist(p, solid); the segments are part of the solid, so their distances are at least dist(p, solid) too.  Each term
        is therefore >= dist(p, solid) / sqrt(1 + G^2) with G the largest |z'| over [0, d/2] — taken from 4097 samples, the
        edge included, with 5 % on top.  A profile whose conic term runs out of its domain inside the aperture (NaN) gets none."""
        return _profile_slope_bound(self.radius, self.conic_constant, self.coefficients, self.diameter)

    def _local_bound(self):
        # the closed 2D perimeter lives in |r| <= d/2, z between 0, the edge sag and the extreme sag
        zs = [0.0, self.edge_sag_value(), self.max_sag[0]]
        lo, hi = min(zs), max(zs)
        return [0, (lo + hi) / 2, 0], math.hypot((hi - lo) / 2, self.diameter / 2)


class ConvexAsphericalSurfaceSDF(_AsphericalSurfaceSDF):  # AsphericalLensSDF.jl:19-48
    kind = K_ASPH_CONVEX

    @property
    def thickness(self):
        sag = self.edge_sag_value()
        return self.max_sag[0] if (self.max_sag[0] > 0 and sag < 0) else abs(sag)


class ConcaveAsphericalSurfaceSDF(_AsphericalSurfaceSDF):  # AsphericalLensSDF.jl:85-121
    kind = K_ASPH_CONCAVE

    def __init__(self, coefficients, radius, conic_constant, diameter, mechanical_diameter=None):
        super().__init__(coefficients, radius, conic_constant, diameter)
        self.mechanical_diameter = float(diameter if mechanical_diameter is None else mechanical_diameter)

    @property
    def thickness(self):
        sag = self.edge_sag_value()
        return abs(sag) if (self.max_sag[0] > 0 and sag < 0) else 0.0


class EvenAsphericalSurface:  # AsphericalLensSDF.jl:426-497
    def __init__(self, radius, diameter, conic_constant, coefficients, mechanical_diameter=None):
        self.radius = float(radius)
        self.diameter = float(diameter)
        self.mechanical_diameter = float(diameter if mechanical_diameter is None else mechanical_diameter)
        self.conic_constant = float(conic_constant)
        self.coefficients = [float(x) for x in coefficients]


def PlanoConvexAsphericalLensSDF(r, l, d, k, coeffs):  # AsphericalLensSDF.jl:364-376
    s = aspheric_equation(d / 2, 1 / r, k, coeffs)
    front = ConvexAsphericalSurfaceSDF(coeffs, r, k, d)
    back = CylinderSDF(d / 2, (l - abs(s)) / 2)
    translate3d(front, [0, -_sign(r) * (l / 2 + abs(s) / 2), 0])
    return front + back


# ------------------------------------------------------------------- cylinder lenses (CylindricalSDF.jl)
class ConvexCylinderSDF(AbstractSDF):  # CylindricalSDF.jl:25-60
    kind = K_CYL_CONVEX

    def __init__(self, radius, diameter, height):
        super().__init__()
        self.radius, self.diameter, self.height = float(radius), float(diameter), float(height)
        xrotate3d(self, math.pi / 2)
        translate3d(self, [0, radius, 0])

    @property
    def thickness(self):
        return abs(la.sag(self.radius, self.diameter))

    def params(self):
        return [self.radius, self.diameter, self.height]

    def _local_bound(self):
        # local frame: extrusion along x (|x| <= h/2), cut disk of radius r in (y, z) with z >= sqrt(r^2 - (d/2)^2)
        hc = math.sqrt(self.radius ** 2 - (self.diameter / 2) ** 2)
        zc = (hc + self.radius) / 2
        return [0, 0, zc], math.sqrt((self.height / 2) ** 2 + (self.diameter / 2) ** 2 + ((self.radius - hc) / 2) ** 2)


class ConcaveCylinderSDF(AbstractSDF):  # CylindricalSDF.jl:92-139
    kind = K_CYL_CONCAVE
    thickness = 0.0

    def __init__(self, radius, diameter, height):
        super().__init__()
        self.radius, self.diameter, self.height = float(radius), float(diameter), float(height)

    def params(self):
        return [self.radius, self.diameter, self.height]

    def _local_bound(self):
        sg = la.sag(abs(self.radius), self.diameter)
        yc = sg / 2 * _sign(self.radius)
        return [0, yc, 0], math.sqrt((self.height / 2) ** 2 + (sg / 2) ** 2 + (self.diameter / 2) ** 2)


class CylindricalSurface:  # CylindricalSDF.jl:160-215
    def __init__(self, radius, diameter, height, mechanical_diameter=None):
        self.radius, self.diameter, self.height = float(radius), float(diameter), float(height)
        self.mechanical_diameter = float(diameter if mechanical_diameter is None else mechanical_diameter)


class RectangularFlatSurface:  # CylindricalSDF.jl:224-229
    def __init__(self, size):
        self.size = self.diameter = self.mechanical_diameter = float(size)
        self.radius = math.inf


def _enclose(spheres):
    """Conservative sphere around a list of (centre, radius)."""
    c, r = np.array(spheres[0][0], dtype=np.float64), float(spheres[0][1])
    for c2, r2 in spheres[1:]:
        c2 = np.asarray(c2, dtype=np.float64)
        d = float(np.linalg.norm(c2 - c))
        if d + r2 <= r:
            continue
        if d + r <= r2:
            c, r = c2.copy(), float(r2)
            continue
        nr = (d + r + r2) / 2
        c = c + (c2 - c) * ((nr - r) / d)
        r = nr
    return c, r


class UnionSDF(AbstractSDF):  # UnionSDF.jl:22-91
    kind = K_UNION

    def __init__(self, *sdfs):
        super().__init__()
        self.sdfs = list(sdfs)

    @property
    def thickness(self):  # UnionSDF.jl:34-42
        return sum(getattr(s, "thickness") for s in self.sdfs if hasattr(s, "thickness"))

    def _translate3d(self, offset):  # UnionSDF.jl:63-67
        off = la.vec3(offset)
        self.pos = self.pos + off
        for s in self.sdfs:
            s._translate3d(off)

    def _rotate3d(self, axis, theta):  # UnionSDF.jl:69-82
        R = la.rotate3d(axis, theta)
        self._set_orientation(R @ self.dir)
        for s in self.sdfs:
            s._rotate3d(axis, theta)
            v = s.pos - self.pos
            v = (R @ v) - v
            s._translate3d(v)

    def world_bound(self):
        return _enclose([s.world_bound() for s in self.sdfs])


class MeniscusLensSDF(AbstractSDF):  # MeniscusLensSDF.jl:19-46
    kind = K_MENISCUS

    def __init__(self, convex, cylinder, concave, thickness):
        super().__init__()
        self.convex, self.cylinder, self.concave = convex, cylinder, concave
        self.thickness = float(thickness)

    @property
    def diameter(self):
        return self.cylinder.diameter

    @property
    def sdfs(self):
        return [self.convex, self.cylinder, self.concave]

    def _local_bound(self):
        return _enclose([self.convex.world_bound(), self.cylinder.world_bound()])


# --- kinematic free functions (reference names without the `!`)
def translate3d(obj, offset):
    obj._translate3d(offset)


def translate_to3d(obj, target):  # AbstractShape.jl:59-63
    translate3d(obj, la.vec3(target) - position(obj))


def rotate3d(obj, axis, theta):
    obj._rotate3d(la.vec3(axis), theta)


def xrotate3d(obj, theta):
    rotate3d(obj, [1.0, 0.0, 0.0], theta)


def yrotate3d(obj, theta):
    rotate3d(obj, [0.0, 1.0, 0.0], theta)


def zrotate3d(obj, theta):
    rotate3d(obj, [0.0, 0.0, 1.0], theta)


def align3d(obj, target_axis):
    obj._align3d(la.vec3(target_axis))


def reset_translation3d(obj):
    obj._reset_translation3d()


def reset_rotation3d(obj):
    obj._reset_rotation3d()


def position(obj):
    return obj.position() if callable(getattr(obj, "position", None)) else obj.pos


def orientation(obj):
    return obj.orientation() if callable(getattr(obj, "orientation", None)) else obj.dir


def thickness(obj):
    t = obj.thickness
    return t() if callable(t) else t


def set_new_origin3d(obj):
    (obj.shape if hasattr(obj, "shape") and not isinstance(obj, AbstractShape) else obj).set_new_origin3d()


# ------------------------------------------------------------------- lens shape builders
def ThinLensSDF(r1, r2, d=la.inch):  # SphericalLensSDF.jl:245-253
    front = ConvexSphericalSurfaceSDF(r1, d)
    back = ConvexSphericalSurfaceSDF(r2, d)
    translate3d(back, [0, front.thickness + back.thickness, 0])
    zrotate3d(back, math.pi)
    return front + back


def BiConvexLensSDF(r1, r2, l, d=la.inch):  # SphericalLensSDF.jl:267-282
    la.check_sag(r1, d)
    la.check_sag(r2, d)
    l = l - (la.sag(r1, d) + la.sag(r2, d))
    front = ConvexSphericalSurfaceSDF(r1, d)
    back = ConvexSphericalSurfaceSDF(r2, d)
    mid = PlanoSurfaceSDF(l, d)
    translate3d(mid, [0, front.thickness, 0])
    zrotate3d(back, math.pi)
    translate3d(back, [0, front.thickness + mid.thickness + back.thickness, 0])
    return front + mid + back


def BiConcaveLensSDF(r1, r2, l, d=la.inch, md=None):  # SphericalLensSDF.jl:284-325
    front = ConcaveSphericalSurfaceSDF(r1, d)
    back = ConcaveSphericalSurfaceSDF(r2, d)
    mid = PlanoSurfaceSDF(l, d)
    zrotate3d(back, math.pi)
    translate3d(back, [0, mid.thickness, 0])
    shape = front + mid + back
    if md is not None:
        if md <= d:
            raise ValueError("Mech. diameter must be larger than lens diameter!")
        l0 = l + shape.sdfs[0].sag + shape.sdfs[2].sag
        ring = RingSDF(d / 2, (md - d) / 2, l0)
        translate3d(ring, [0, l / 2, 0])
        shape = shape + ring
    return shape


def PlanoConvexLensSDF(r, l, d=la.inch):  # SphericalLensSDF.jl:338-347
    l = l - la.sag(r, d)
    front = ConvexSphericalSurfaceSDF(r, d)
    back = PlanoSurfaceSDF(l, d)
    translate3d(back, [0, front.thickness, 0])
    return front + back


def PlanoConcaveLensSDF(r, l, d=la.inch, md=None):  # SphericalLensSDF.jl:349-381
    front = PlanoSurfaceSDF(l, d)
    back = ConcaveSphericalSurfaceSDF(r, d)
    zrotate3d(back, math.pi)
    translate3d(back, [0, front.thickness, 0])
    shape = front + back
    if md is not None:
        if md <= d:
            raise ValueError("Mech. diameter must be larger than lens diameter!")
        _l = l + shape.sdfs[1].sag
        ring = RingSDF(d / 2, (md - d) / 2, _l)
        translate3d(ring, [0, _l / 2, 0])
        shape = shape + ring
    return shape


# ----------------------------------------------------------------------- surface specs
class CircularFlatSurface:  # AbstractSurface.jl:95-104
    def __init__(self, diameter):
        self.diameter = float(diameter)
        self.mechanical_diameter = float(diameter)
        self.radius = math.inf


class SphericalSurface:  # SphericalLensSDF.jl:397-454
    def __init__(self, radius, diameter, mechanical_diameter=None):
        self.radius = float(radius)
        self.diameter = float(diameter)
        self.mechanical_diameter = float(diameter if mechanical_diameter is None else mechanical_diameter)


def _surface_sdf(s, orient):
    """sdf(surface, orientation) SphericalLensSDF.jl:423-454 / AbstractSurface.jl:104."""
    if isinstance(s, (CircularFlatSurface, RectangularFlatSurface)) or math.isinf(s.radius):
        return None
    if isinstance(s, EvenAsphericalSurface):  # AsphericalLensSDF.jl:473-497 (no pi rotation: the sign of r carries it)
        args = (s.coefficients, s.radius, s.conic_constant, s.diameter)
        if orient == "forward":
            return ConvexAsphericalSurfaceSDF(*args) if s.radius > 0 else ConcaveAsphericalSurfaceSDF(*args)
        if orient == "backward":
            return ConcaveAsphericalSurfaceSDF(*args) if s.radius > 0 else ConvexAsphericalSurfaceSDF(*args)
        raise ValueError("aspheric meniscus lenses are not supported by the reference either")
    if isinstance(s, AcylindricalSurface):
        return _acyl_surface_sdf(s, orient)
    if isinstance(s, CylindricalSurface):  # CylindricalSDF.jl:187-215
        if orient == "forward":
            return ConvexCylinderSDF(s.radius, s.diameter, s.height) if s.radius > 0 else ConcaveCylinderSDF(s.radius, s.diameter, s.height)
        if orient == "backward":
            return ConcaveCylinderSDF(s.radius, s.diameter, s.height) if s.radius > 0 else ConvexCylinderSDF(-s.radius, s.diameter, s.height)
        raise ValueError(orient)
    if orient == "forward":
        return ConvexSphericalSurfaceSDF(s.radius, s.diameter) if s.radius > 0 else ConcaveSphericalSurfaceSDF(abs(s.radius), s.diameter)
    if orient == "backward":
        back = ConcaveSphericalSurfaceSDF(s.radius, s.diameter) if s.radius > 0 else ConvexSphericalSurfaceSDF(abs(s.radius), s.diameter)
        zrotate3d(back, math.pi)
        return back
    if orient == "forward_left_meniscus":
        return ConvexSphericalSurfaceSDF(s.radius, s.diameter)
    if orient == "backward_left_meniscus":
        return SphereSDF(s.radius)
    if orient == "forward_right_meniscus":
        return SphereSDF(abs(s.radius))
    if orient == "backward_right_meniscus":
        return ConvexSphericalSurfaceSDF(abs(s.radius), s.diameter)
    raise ValueError(orient)


def _edge_sag(surface, sd):  # SphericalLensSDF.jl:421, AsphericalLensSDF.jl:462-471, CylindricalSDF.jl:184-185
    if isinstance(surface, EvenAsphericalSurface):
        return aspheric_equation(surface.diameter / 2, 1 / surface.radius, surface.conic_constant, surface.coefficients)
    if isinstance(surface, AcylindricalSurface):
        return _acyl_edge_sag(surface)
    if isinstance(surface, CylindricalSurface):
        return sd.thickness
    return sd.sag


def _sign(x):
    return (x > 0) - (x < 0)


def meniscus_lens_sdf(front_surface, front, back_surface, back, center_thickness):  # MeniscusLensSDF.jl:122-189
    r1, r2 = front_surface.radius, back_surface.radius
    if _sign(r1) == _sign(r2) and _sign(r1) > 0:
        orient = "left"
    elif _sign(r1) == _sign(r2) and _sign(r1) < 0:
        orient = "right"
    else:
        raise ValueError("Invalid sign combination for r1 and r2")
    convex_sag = _edge_sag(front_surface, front)
    concave_sag = _edge_sag(back_surface, back)
    cylinder_l = center_thickness - convex_sag + concave_sag
    if cylinder_l <= 0:
        raise ValueError("Lens parameters lead to zero lens edge thickness")
    if orient == "left":
        front = _surface_sdf(front_surface, "forward_left_meniscus")
        back = _surface_sdf(back_surface, "backward_left_meniscus")
    else:
        front = _surface_sdf(front_surface, "forward_right_meniscus")
        back = _surface_sdf(back_surface, "backward_right_meniscus")
    d_mid = min(front_surface.diameter, back_surface.diameter)
    cylinder = PlanoSurfaceSDF(cylinder_l, d_mid)
    if orient == "left":
        translate3d(cylinder, [0, front.thickness, 0])
        translate3d(back, [0, back_surface.radius + center_thickness, 0])
        convex_shape, concave_shape = front, back
    else:
        translate3d(back, [0, -abs(front_surface.radius), 0])
        translate3d(cylinder, [0, -concave_sag, 0])
        zrotate3d(front, math.pi)
        translate3d(front, [0, cylinder.thickness - concave_sag + convex_sag, 0])
        convex_shape, concave_shape = back, front
    return MeniscusLensSDF(convex_shape, cylinder, concave_shape, center_thickness)


def _cyl_lens_shape(front_surface, back_surface, center_thickness):
    """Lens(front::AbstractCylindricalSurface, back::AbstractCylindricalSurface, t, n): Lenses.jl:331-388, :432-446."""
    if isinstance(front_surface, RectangularFlatSurface) and isinstance(back_surface, RectangularFlatSurface):
        d_mid = min(front_surface.diameter, back_surface.diameter)
        mid = BoxSDF(d_mid, center_thickness, d_mid)
        translate3d(mid, [0, center_thickness / 2, 0])
        return mid
    l0 = center_thickness
    front = _surface_sdf(front_surface, "forward")
    l0 -= 0.0 if front is None else front.thickness
    back = _surface_sdf(back_surface, "backward")
    l0 -= 0.0 if back is None else back.thickness
    # cylindric_lens_outer_parameters Lenses.jl:401-430
    f, b = front_surface, back_surface
    if isinstance(f, RectangularFlatSurface):
        f, b = b, f
    if isinstance(b, RectangularFlatSurface):
        d_mid, md_mid, h = f.diameter, f.mechanical_diameter, f.height
    else:
        if f.height != b.height:
            raise ValueError("height of front and back surface have to match for cylindric lenses")
        d_mid, md_mid, h = min(f.diameter, b.diameter), max(f.mechanical_diameter, b.mechanical_diameter), f.height
    if l0 <= 0:
        raise ValueError("Lens parameters lead to a box section length of <= 0")
    mid = BoxSDF(h, l0, d_mid)
    translate3d(mid, [0, l0 / 2, 0])
    if front is not None:
        translate3d(mid, [0, front.thickness, 0])
        mid = mid + front
    if back is not None:
        translate3d(back, [0, mid.thickness + back.thickness, 0])
        mid = mid + back
    shape = mid
    if md_mid > d_mid:
        ring_thickness = mid.thickness
        ring_center = mid.pos[1] + ring_thickness / 2
        if front is not None:
            sg = _edge_sag(front_surface, front)
            ring_thickness -= sg
            ring_center += sg / 2
        if back is not None:
            sg = _edge_sag(back_surface, back)
            ring_thickness += sg
            ring_center += sg / 2
        ring = RingSDF(d_mid / 2, (md_mid - d_mid) / 2, ring_thickness)
        translate3d(ring, [0, ring_center, 0])
        shape = shape + ring
    return shape


def lens_shape_from_surfaces(front_surface, back_surface, center_thickness):
    """Shape part of Lens(front_surface, back_surface, center_thickness, n): Lenses.jl:176-311."""
    if isinstance(front_surface, (CylindricalSurface, RectangularFlatSurface)) or isinstance(back_surface, (CylindricalSurface, RectangularFlatSurface)):
        return _cyl_lens_shape(front_surface, back_surface, center_thickness)
    if isinstance(front_surface, CircularFlatSurface) and isinstance(back_surface, CircularFlatSurface):
        return PlanoSurfaceSDF(center_thickness, min(front_surface.diameter, back_surface.diameter))  # Lenses.jl:304-311
    d_mid = min(front_surface.diameter, back_surface.diameter)
    md_mid = max(front_surface.mechanical_diameter, back_surface.mechanical_diameter)
    l0 = center_thickness
    front = _surface_sdf(front_surface, "forward")
    l0 -= 0.0 if front is None else front.thickness
    back = _surface_sdf(back_surface, "backward")
    l0 -= 0.0 if back is None else back.thickness
    if l0 <= 0:
        if _sign(front_surface.radius) == _sign(back_surface.radius):
            shape = meniscus_lens_sdf(front_surface, front, back_surface, back, center_thickness)
            if md_mid > d_mid:
                _t = shape.cylinder.thickness
                pos = shape.cylinder.pos
                ring = RingSDF(d_mid / 2, (md_mid - d_mid) / 2, _t)
                translate3d(ring, [0, pos[1] + _t / 2, 0])
                shape = shape + ring
            return shape
        raise ValueError("Lens parameters lead to cylinder section length of <= 0, use ThinLens instead.")
    mid = PlanoSurfaceSDF(l0, d_mid)
    plano = mid
    if front is not None:
        translate3d(mid, [0, front.thickness, 0])
        mid = mid + front
    if back is not None:
        translate3d(back, [0, mid.thickness + back.thickness, 0])
        mid = mid + back
    shape = mid
    d_front, d_back = front_surface.diameter, back_surface.diameter
    d_min, d_max = min(d_front, d_back), max(d_front, d_back)
    if md_mid < d_min:
        return shape
    if d_front != d_back:
        if d_back > d_front:
            lev = l0
            if front is not None:
                s_front = _edge_sag(front_surface, front)
                if s_front < 0:
                    lev += abs(s_front) + front.thickness
            ring = RingSDF(d_front / 2, (d_back - d_front) / 2, lev)
            translate3d(ring, [0, _edge_sag(front_surface, front) + lev / 2, 0])
            shape = shape + ring
        else:
            lev = l0
            if back is not None:
                s_back = _edge_sag(back_surface, back)
                if (s_back - back.thickness) > 0:
                    lev += abs(s_back) + back.thickness
            ring = RingSDF(d_back / 2, (d_front - d_back) / 2, lev)
            translate3d(ring, [0, front.thickness + lev / 2, 0])
            shape = shape + ring
    if md_mid > d_max:
        # `mid` here is the UnionSDF (position stays at the origin), as in the reference
        outer_thickness = mid.thickness
        outer_center = mid.pos[1] + outer_thickness / 2
        if front is not None:
            s_front = _edge_sag(front_surface, front)
            outer_thickness -= s_front
            outer_center += s_front / 2
        if back is not None:
            s_back = _edge_sag(back_surface, back)
            outer_thickness += s_back
            outer_center += s_back / 2
        ring = RingSDF(d_max / 2, (md_mid - d_max) / 2, outer_thickness)
        translate3d(ring, [0, outer_center, 0])
        shape = shape + ring
    return shape


# acylinders live in their own module (AcylindricalSDF.jl); import last so they can subclass the types above
import sys as _sys  # noqa: E402

from . import acylinders as _acyl  # noqa: E402

K_ACYL_CONVEX, K_ACYL_CONCAVE = _acyl.K_ACYL_CONVEX, _acyl.K_ACYL_CONCAVE
AconvexCylinderSDF, AconcaveCylinderSDF, AcylindricalSurface, _acyl_surface_sdf, _acyl_edge_sag = _acyl.make(_sys.modules[__name__])
