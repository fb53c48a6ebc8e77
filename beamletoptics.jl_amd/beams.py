"""Host-side mirror of the reference's beam containers (the batch axis of the hot path).

Ray / PolarizedRay / Beam / GaussianBeamlet objects here are plain Python records that the
solver fills from the engine's result tables; bundle generators build the SoA input planes.
Citations relative to /root/reference/src.
"""
import math

import numpy as np

from . import linalg as la

BEAM_RAY, BEAM_POLARIZED, BEAM_GAUSSIAN = 0, 1, 2


class Intersection:
    """AbstractTypes/AbstractRay.jl:13-18"""

    __slots__ = ("object", "shape", "t", "n")

    def __init__(self, t, n, object=None, shape=None):
        self.t, self.n, self.object, self.shape = t, n, object, shape

    def __len__(self):
        raise TypeError("use .t")


class Ray:
    """Rays.jl:14-42 (public ctor normalises dir, n = 1)."""

    __slots__ = ("pos", "dir", "intersection", "lam", "n")

    def __init__(self, pos, dir, lam=1000e-9, n=1.0, _raw=False):
        self.pos = la.vec3(pos)
        self.dir = la.vec3(dir) if _raw else la.normalize(la.vec3(dir))
        self.intersection = None
        self.lam = float(lam)
        self.n = float(n)

    def length(self):
        return math.inf if self.intersection is None else self.intersection.t

    def optical_path_length(self):
        return math.inf if self.intersection is None else self.intersection.t * self.n


def electric_field_from_intensity(I, Z=la.Z_vacuum):
    """OpticUtils.jl:105 with phi = 0."""
    return math.sqrt(2 * I * Z)


class PolarizedRay(Ray):
    """PolarizedRays.jl:37-95"""

    __slots__ = ("E0",)

    def __init__(self, pos, dir, lam=1000e-9, E0=None, n=1.0, _raw=False):
        super().__init__(pos, dir, lam, n, _raw)
        E0 = [electric_field_from_intensity(1), 0, 0] if E0 is None else E0
        self.E0 = np.asarray(E0, dtype=np.complex128).reshape(3).copy()
        if abs(np.dot(self.dir, self.E0)) > 1e-14:  # PolarizedRays.jl:54-56
            raise ValueError("Ray dir. and E0 must be orthogonal.")


class Beam:
    """Beam.jl:13-25"""

    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], Ray):
            ray = args[0]
        elif len(args) == 3:
            ray = Ray(*args)
        elif len(args) == 4:
            ray = PolarizedRay(*args)
        else:
            raise TypeError("Beam(ray) | Beam(pos, dir, lam) | Beam(pos, dir, lam, E0)")
        self.rays = [ray]
        self.parent = None
        self.children = []
        self.status = 0

    @property
    def kind(self):
        return BEAM_POLARIZED if isinstance(self.rays[0], PolarizedRay) else BEAM_RAY

    def length(self):  # Beam.jl:125-169
        l0 = 0.0 if self.parent is None else self.parent.length()
        l = 0.0
        for r in self.rays:
            if r.intersection is None:
                break
            l += r.intersection.t
        return l + l0

    def optical_path_length(self):  # Beam.jl:137-149
        l0 = 0.0 if self.parent is None else self.parent.optical_path_length()
        for r in self.rays:
            if r.intersection is None:
                break
            l0 += r.optical_path_length()
        return l0


class GaussianBeamlet:
    """Gaussian.jl:33-59, ctor :215-256."""

    kind = BEAM_GAUSSIAN

    def __init__(self, position, direction, lam=1e-6, w0=1e-3, M2=1, P0=1e-3, z0=0, support=None, _parts=None):
        self.parent = None
        self.children = []
        self.status = 0
        if _parts is not None:
            self.chief, self.waist, self.divergence, self.lam, self.w0, self.E0 = _parts
            return
        position = la.vec3(position)
        d = la.normalize(la.vec3(direction))
        s1 = la.perpendicular(d) if support is None else la.vec3(support)
        s1 = la.normalize(s1)
        tan_t = math.tan(M2 * lam / (math.pi * w0))  # divergence_angle OpticUtils.jl:63
        wst = Ray(position + s1 * w0, d, lam)
        div_dir = la.normalize(d + s1 * tan_t)
        dz = -z0 * tan_t
        div = Ray(position + s1 * dz, div_dir, lam)
        chf = Ray(position, d, lam)
        I0 = 2 * P0 / (math.pi * w0 ** 2)
        self.chief, self.waist, self.divergence = Beam(chf), Beam(wst), Beam(div)
        self.lam = float(lam)
        self.w0 = float(w0)
        self.E0 = complex(electric_field_from_intensity(I0))

    @property
    def rays(self):
        return self.chief.rays

    def length(self):
        return self.chief.length()

    def optical_path_length(self):  # Gaussian.jl:102
        return self.chief.optical_path_length()


def point_on_beam(beam, t):
    """Beam.jl:177-205: point at geometric distance t along the beam (parents included) and its 1-based segment index."""
    temp = 0.0 if beam.parent is None else beam.parent.length()
    num = len(beam.rays)
    for index, ray in enumerate(beam.rays, start=1):
        if index == num:
            break
        temp += ray.length()
        if t < temp:
            b = temp - t
            return ray.pos + (ray.length() - b) * ray.dir, index
    ray = beam.rays[-1]
    return ray.pos + (t - temp) * ray.dir, num


def gauss_parameters(gauss, z):
    """Gaussian.jl:298-353: (w, R, psi, w0) of a traced GaussianBeamlet at path length z (Arnaud / DeJager complex rays)."""
    p0, index = point_on_beam(gauss.chief, z)
    chief, div, waist = gauss.chief.rays[index - 1], gauss.divergence.rays[index - 1], gauss.waist.rays[index - 1]

    def height_slope(r):
        denom = np.dot(chief.dir, r.dir)
        il = np.dot(p0 - r.pos, chief.dir) / denom if abs(denom) > 1e-6 else math.nan
        y0 = r.pos + il * r.dir - p0
        y = math.sqrt(np.dot(y0, y0))
        with np.errstate(all="ignore"):
            y0 = y0 / y
            arg = max(-1.0, min(1.0, np.dot(y0, r.dir) / (np.linalg.norm(y0) * np.linalg.norm(r.dir)))) if y > 0 else math.nan
        return y, math.tan(math.pi / 2 - math.acos(arg)) if arg == arg else math.nan

    y_d, m_d = height_slope(div)
    y_w, m_w = height_slope(waist)
    n = chief.n
    H = abs(n * (y_w * m_d - y_d * m_w))
    lam = gauss.lam
    if not abs(H - lam / math.pi) <= 1e-6:
        H = lam / math.pi
    E_kt = y_d * m_d + y_w * m_w
    F_kt = math.sqrt(m_d ** 2 + m_w ** 2)
    w = math.sqrt(y_d ** 2 + y_w ** 2)
    with np.errstate(all="ignore"):
        R = np.float64(E_kt) / np.float64(w ** 2)
        zz = np.float64(E_kt) / np.float64(F_kt ** 2)
        psi = -np.arctan2(1.0, np.sqrt(1 / (R * zz) - 1))
        w0 = np.float64(H) / np.float64(n * F_kt)
    R = 0.0 if np.isnan(R) else float(R)
    psi = 0.0 if np.isnan(psi) else float(psi)
    w0 = w if np.isnan(w0) else float(w0)
    if R < 0:
        psi = -psi
    return w, R, psi, w0


class BeamGroup:
    """AbstractBeamGroup (AbstractBeam.jl:99-124): `beams` is the batch."""

    def __init__(self, beams, **meta):
        self.beams = list(beams)
        self.__dict__.update(meta)


def CollimatedSource(pos, dir, diameter, lam=1e-6, num_rings=10, num_rays=None, b1=None):
    """BeamGroups.jl:152-195.  `b1` fixes the basis vector the reference draws at random (:169)."""
    num_rays = 100 * num_rings if num_rays is None else num_rays
    if num_rays < num_rings * 20:
        raise ValueError("No. of rays should be atleast 20x no. of rings")
    pos = la.vec3(pos)
    dir = la.vec3(dir)
    beams = [Beam(Ray(pos, dir, lam))]
    num_rays -= 1
    b1 = la.perpendicular(dir) if b1 is None else la.vec3(b1)
    radii = np.linspace(0, diameter / 2, num_rings)[1:]
    circm = radii * 2 * math.pi
    ds = circm.sum() / num_rays
    n_rays = np.round(circm / ds).astype(int)
    n_rays[-1] += num_rays - n_rays.sum()
    for r, numEl in zip(radii, n_rays):
        if numEl == 0:
            continue
        Rm = la.rotate3d(dir, 2 * math.pi / numEl)
        helper = b1 * r
        for _ in range(numEl):
            beams.append(Beam(pos + helper, dir, lam))
            helper = Rm @ helper
    return BeamGroup(beams, diameter=float(diameter))


def PointSource(pos, dir, theta, lam=1e-6, num_rings=10, num_rays=None, b1=None):
    """BeamGroups.jl:51-100."""
    num_rays = 100 * num_rings if num_rays is None else num_rays
    if num_rays < num_rings * 20:
        raise ValueError("No. of rays should be atleast 20x no. of rings")
    if theta >= math.pi:
        raise ValueError("Point source opening half-angle must be <= pi")
    pos = la.vec3(pos)
    dir = la.normalize(la.vec3(dir))
    b1 = la.perpendicular(dir) if b1 is None else la.vec3(b1)
    b2 = la.normal3d(dir, b1)
    th = np.linspace(0, theta, num_rings)
    step = th[1] - th[0]
    beams = [Beam(Ray(pos, dir, lam))]
    num_rays -= 1
    ndirs = [la.rotate3d(b2, step * (i + 1)) @ dir for i in range(num_rings - 1)]
    circm = np.array([np.linalg.norm(nd - np.dot(nd, dir) * dir) for nd in ndirs]) * 2 * math.pi
    ds = circm.sum() / num_rays
    n_rays = np.round(circm / ds).astype(int)
    n_rays[-1] += num_rays - n_rays.sum()
    for nd, numEl in zip(ndirs, n_rays):
        if numEl == 0:
            continue
        Rm = la.rotate3d(dir, 2 * math.pi / numEl)
        c = nd
        for _ in range(numEl):
            beams.append(Beam(pos, c, lam))
            c = Rm @ c
    return BeamGroup(beams, NA=math.sin(theta))


def UniformDiscSource(pos, dir, diameter, lam=1e-6, num_rays=1000, e1=None):
    """BeamGroups.jl:222-245 (Fibonacci disc).  `e1` fixes the basis the reference draws at random."""
    pos = la.vec3(pos)
    dir = la.vec3(dir)
    e1 = la.perpendicular(dir) if e1 is None else la.vec3(e1)
    e2 = la.normalize(np.cross(dir, e1))
    R = diameter / 2
    phi0 = 2 * math.pi / (1 + math.sqrt(5))
    beams = []
    for k in range(num_rays):
        rho = math.sqrt((k + 0.5) / num_rays)
        phi = k * phi0
        r = R * rho
        x = r * math.cos(phi) * e1 + r * math.sin(phi) * e2
        beams.append(Beam(pos + x, dir, lam))
    return BeamGroup(beams, diameter=float(diameter))


# ------------------------------------------------------------------ SoA bundle (large batches)
class RayBundle:
    """Structure-of-arrays batch for large N (no per-ray Python objects).

    planes: (n_planes, N) float64 in the order documented in include/bmo.h; kind: BEAM_*.
    """

    def __init__(self, kind, planes):
        self.kind = int(kind)
        self.planes = np.ascontiguousarray(planes, dtype=np.float64)

    @property
    def n(self):
        return self.planes.shape[1]

    @property
    def lambdas(self):
        return self.planes[{0: 6, 1: 6, 2: 18}[self.kind]]

    @staticmethod
    def from_beams(beams):
        b0 = beams[0]
        kind = b0.kind
        n = len(beams)
        if kind == BEAM_GAUSSIAN:
            P = np.zeros((25, n))
            for i, g in enumerate(beams):
                for base, bm in ((0, g.chief), (6, g.waist), (12, g.divergence)):
                    r = bm.rays[0]
                    P[base:base + 3, i] = r.pos
                    P[base + 3:base + 6, i] = r.dir
                P[18, i] = g.lam
                P[19, i] = g.chief.rays[0].n
                P[20, i] = g.w0
                P[21, i] = g.E0.real
                P[22, i] = g.E0.imag
            return RayBundle(kind, P)
        npl = 8 if kind == BEAM_RAY else 14
        P = np.zeros((npl, n))
        for i, b in enumerate(beams):
            r = b.rays[0]
            P[0:3, i] = r.pos
            P[3:6, i] = r.dir
            P[6, i] = r.lam
            P[7, i] = r.n
            if kind == BEAM_POLARIZED:
                P[8:14:2, i] = r.E0.real
                P[9:14:2, i] = r.E0.imag
        return RayBundle(kind, P)

    @staticmethod
    def rays(pos, dir, lam, n=1.0):
        """pos, dir: (N,3); dir is normalised like the Ray ctor (x ./ norm(x))."""
        pos = np.asarray(pos, dtype=np.float64)
        dir = np.asarray(dir, dtype=np.float64)
        nrm = np.sqrt((dir[:, 0] * dir[:, 0] + dir[:, 1] * dir[:, 1]) + dir[:, 2] * dir[:, 2])
        dir = dir / nrm[:, None]
        N = pos.shape[0]
        P = np.zeros((8, N))
        P[0:3] = pos.T
        P[3:6] = dir.T
        P[6] = lam
        P[7] = n
        return RayBundle(BEAM_RAY, P)
