"""Host-side mirror of the reference's optical components (constructors + kinematics only).

The optical *interaction* of every component (interact3d) runs inside the HIP engine; the
classes here only carry geometry and parameters into the flat object table.  Citations are
relative to /root/reference/src/OpticalComponents.
"""
import math

import numpy as np

from . import linalg as la
from . import shapes as sh

# object kinds (include/bmo.h enum bmo_object_kind)
(O_MIRROR, O_REFRACTIVE, O_DOUBLET, O_THIN_BS, O_PLATE_BS, O_CUBE_BS, O_SPOT, O_PSF, O_INTERSECTABLE, O_NONINTERACTABLE,
 O_POLARIZER, O_PHOTODETECTOR) = range(12)


# ------------------------------------------------------------------ refractive indices
class DiscreteRefractiveIndex:
    """Utils/RefractiveIndexUtils.jl:8-31: exact-key lookup, KeyError otherwise."""

    def __init__(self, lambdas, ns):
        if len(lambdas) != len(ns):  # RefractiveIndexUtils.jl:23-25 (ArgumentError)
            raise ValueError("Number of wavelengths must match number of ref. indices")
        self.data = {float(l): float(n) for l, n in zip(lambdas, ns)}

    def __call__(self, lam):
        return self.data[float(lam)]


class SellmeierEquation:
    """Utils/RefractiveIndexUtils.jl:82-98."""

    def __init__(self, B1, B2, B3, C1, C2, C3):
        self.c = (B1, B2, B3, C1, C2, C3)

    def __call__(self, lam):
        B1, B2, B3, C1, C2, C3 = self.c
        lam = lam * 1e6
        n2 = 1 + (B1 * lam ** 2) / (lam ** 2 - C1) + (B2 * lam ** 2) / (lam ** 2 - C2) + (B3 * lam ** 2) / (lam ** 2 - C3)
        return math.sqrt(n2)


def _as_index(n):
    if callable(n):
        return n
    val = float(n)
    return lambda lam: val


# ---------------------------------------------------------------------- object bases
class AbstractObject:
    """AbstractTypes/AbstractObject.jl:36 with the SingleShape trait (AbstractShapeTrait.jl:31-52)."""

    kind = None

    def __init__(self, shape):
        self.shape = shape

    def parts(self):
        return [self.shape]

    def position(self):
        return self.shape.pos

    def orientation(self):
        return self.shape.dir

    def _translate3d(self, offset):
        self.shape._translate3d(offset)

    def _rotate3d(self, axis, theta):
        self.shape._rotate3d(axis, theta)

    def _align3d(self, axis):
        self.shape._align3d(axis)

    def _reset_translation3d(self):
        self.shape._reset_translation3d()

    def _reset_rotation3d(self):
        self.shape._reset_rotation3d()

    @property
    def thickness(self):
        return sh.thickness(self.shape)


class _MultiShape(AbstractObject):
    """MultiShape trait kinematics, AbstractShapeTrait.jl:54-128."""

    def subparts(self):
        raise NotImplementedError

    def position(self):
        return sh.position(self.subparts()[0])

    def orientation(self):
        return sh.orientation(self.subparts()[0])

    def _set_position(self, pos):
        pass

    def _set_orientation(self, d):
        pass

    def _translate3d(self, offset):
        off = la.vec3(offset)
        self._set_position(self.position() + off)
        for p in self.subparts():
            p._translate3d(off)

    def _rotate3d(self, axis, theta):
        R = la.rotate3d(axis, theta)
        self._set_orientation(R @ self.orientation())
        for p in self.subparts():
            p._rotate3d(axis, theta)
            v = sh.position(p) - self.position()  # position(object) re-read per part (AbstractShapeTrait.jl:122)
            v = (R @ v) - v
            p._translate3d(v)

    def _reset_translation3d(self):  # AbstractShapeTrait.jl:144-150 (sub-part relative translations are kept)
        self._translate3d(-self.position())
        self._set_position(np.zeros(3))

    def _reset_rotation3d(self):  # AbstractShapeTrait.jl:160-173: undo the net rotation about its axis (sub-part rotations are kept)
        R = self.orientation()
        theta = math.acos(min(1.0, max(-1.0, (np.trace(R) - 1) / 2)))
        if theta == 0:
            return
        axis = 1 / (2 * math.sin(theta)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        self._rotate3d(axis, -theta)
        self._set_orientation(np.eye(3))


class ObjectGroup(_MultiShape):
    """ObjectGroups.jl:21-47."""

    kind = None

    def __init__(self, objects):
        self.objects = list(objects)
        self.center = np.zeros(3)
        self.dir = np.eye(3)

    def subparts(self):
        return self.objects

    def position(self):
        return self.center

    def orientation(self):
        return self.dir

    def _set_position(self, pos):
        self.center = np.array(pos, dtype=np.float64)

    def _set_orientation(self, d):
        self.dir = np.array(d, dtype=np.float64)


def leaves(objs):
    """Leaves(system.objects) System.jl:21: depth-first flattening of object groups."""
    out = []
    for o in objs:
        if isinstance(o, ObjectGroup):
            out.extend(leaves(o.objects))
        else:
            out.append(o)
    return out


# ------------------------------------------------------------------------- components
class Mirror(AbstractObject):  # Mirrors.jl:76-78
    kind = O_MIRROR


def SquarePlanoMirror2D(size):  # Mirrors.jl:93-96
    return Mirror(sh.QuadraticFlatMesh(size))


def RectangularPlanoMirror2D(width, height):
    return Mirror(sh.RectangularFlatMesh(width, height))


def RectangularPlanoMirror(width, height, thickness):  # Mirrors.jl:110-119
    shape = sh.CuboidMesh(width, thickness, height)
    sh.translate3d(shape, [-width / 2, 0, -height / 2])
    shape.set_new_origin3d()
    return Mirror(shape)


def SquarePlanoMirror(width, thickness):  # Mirrors.jl:134-136
    return RectangularPlanoMirror(width, width, thickness)


def RoundPlanoMirror(diameter, thickness):  # Mirrors.jl:161-164
    return Mirror(sh.PlanoSurfaceSDF(thickness, diameter))


def ConcaveSphericalMirror(radius, thickness, diameter):  # Mirrors.jl:195-200
    cylinder = sh.PlanoSurfaceSDF(thickness, diameter)
    concave = sh.ConcaveSphericalSurfaceSDF(abs(radius), diameter)
    return Mirror(concave + cylinder)


def RightAnglePrismMirror(leg_length, height):  # Mirrors.jl:226-230
    shape = sh.RightAnglePrismSDF(leg_length, height)
    sh.zrotate3d(shape, math.radians(45 + 180))
    return Mirror(shape)


def Retroreflector(scale):  # Misc.jl:34-49
    return Mirror(sh.RetroMesh(scale))


class Lens(AbstractObject):
    """Lenses.jl:145-155.  Lens(shape, n) or Lens(front_surface, back_surface, thickness, n) (:176-311)."""

    kind = O_REFRACTIVE

    def __init__(self, *args):
        if len(args) == 2:
            shape, n = args
        elif len(args) == 4:
            shape = sh.lens_shape_from_surfaces(args[0], args[1], args[2])
            n = args[3]
        elif len(args) == 3:  # Lens(front_surface, center_thickness, n) :293-302 / :390-399
            flat = sh.RectangularFlatSurface if isinstance(args[0], sh.CylindricalSurface) else sh.CircularFlatSurface
            shape = sh.lens_shape_from_surfaces(args[0], flat(args[0].diameter), args[1])
            n = args[2]
        else:
            raise TypeError("Lens(shape, n) or Lens(front, back, thickness, n)")
        super().__init__(shape)
        self.n = _as_index(n)


class Prism(Lens):  # Prisms.jl:1-16
    pass


def RightAnglePrism(leg_length, height, n):  # Prisms.jl:28-31
    return Prism(sh.RightAnglePrismSDF(leg_length, height), n)


def SphericalLens(r1, r2, l, d=la.inch, n=1.5):  # SphericalLenses.jl:20-32
    if l == 0:
        return ThinLens(r1, r2, d, n)
    return Lens(sh.SphericalSurface(r1, d), sh.SphericalSurface(r2, d), l, n)


def ThinLens(R1, R2, d, n):  # SphericalLenses.jl:40-44
    return Lens(sh.ThinLensSDF(R1, R2, d), n)


class DoubletLens(_MultiShape):  # DoubletLenses.jl:26-38
    kind = O_DOUBLET

    def __init__(self, front, back):
        self.front, self.back = front, back

    def subparts(self):
        return [self.front, self.back]

    def parts(self):
        return [self.front.shape, self.back.shape]

    @property
    def thickness(self):
        return sh.thickness(self.front.shape) + sh.thickness(self.back.shape)


def SphericalDoubletLens(r1, r2, r3, l1, l2, d, n1, n2):  # DoubletLenses.jl:57-64
    front = SphericalLens(r1, r2, l1, d, n1)
    back = SphericalLens(r2, r3, l2, d, n2)
    sh.translate3d(back, [0, sh.thickness(front.shape), 0])
    return DoubletLens(front, back)


class ThinBeamsplitter(AbstractObject):  # Beamsplitters/ThinBeamsplitter.jl:16-67
    kind = O_THIN_BS

    def __init__(self, *args, reflectance=0.5, _shape=None):
        if _shape is None:
            if reflectance >= 1 or abs(reflectance) <= 1.4901161193847656e-08 * abs(reflectance):
                raise ValueError("Splitting ratio in (0, 1)!")
            width = args[0]
            height = args[1] if len(args) > 1 else width
            _shape = sh.RectangularFlatMesh(width, height)
        super().__init__(_shape)
        self.reflectance = math.sqrt(reflectance)
        self.transmittance = math.sqrt(1 - self.reflectance ** 2)


def RoundThinBeamsplitter(diameter, reflectance=0.5):  # ThinBeamsplitter.jl:59-67
    return ThinBeamsplitter(reflectance=reflectance, _shape=sh.CircularFlatMesh(diameter / 2))


class RectangularPlateBeamsplitter(_MultiShape):  # Beamsplitters/PlateBeamsplitter.jl:66-104
    kind = O_PLATE_BS

    def __init__(self, width, height, thickness, n, reflectance=0.5, _round=False):
        if _round:
            self.substrate = Prism(sh.PlanoSurfaceSDF(thickness, width), n)
            self.coating = RoundThinBeamsplitter(width, reflectance=reflectance)
        else:
            self.substrate = Prism(sh.BoxSDF(width, thickness, height), n)
            sh.translate3d(self.substrate, [0, thickness / 2, 0])
            self.coating = ThinBeamsplitter(width, height, reflectance=reflectance)
            sh.zrotate3d(self.coating, math.pi)
        self.n = self.substrate.n

    def subparts(self):  # shape(pbs) = (substrate, coating) PlateBeamsplitter.jl:30
        return [self.substrate, self.coating]

    def position(self):  # PlateBeamsplitter.jl:26-27
        return sh.position(self.coating)

    def orientation(self):
        return sh.orientation(self.substrate)

    def parts(self):
        return [self.substrate.shape, self.coating.shape]


def RoundPlateBeamsplitter(diameter, thickness, n, reflectance=0.5):  # PlateBeamsplitter.jl:141-158
    return RectangularPlateBeamsplitter(diameter, diameter, thickness, n, reflectance=reflectance, _round=True)


class CubeBeamsplitter(_MultiShape):  # Beamsplitters/CubeBeamsplitter.jl:23-61
    kind = O_CUBE_BS

    def __init__(self, leg_length, n, reflectance=0.5):
        self.front = RightAnglePrism(leg_length, leg_length, n)
        self.back = RightAnglePrism(leg_length, leg_length, n)
        self.coating = ThinBeamsplitter(math.sqrt(2) * leg_length, leg_length, reflectance=reflectance)
        sh.zrotate3d(self.back, math.radians(180))
        sh.zrotate3d(self.coating, math.radians(180 - 45))
        self.coating.shape.set_new_origin3d()
        self.n = self.front.n

    def subparts(self):
        return [self.front, self.back, self.coating]

    def parts(self):
        return [self.front.shape, self.back.shape, self.coating.shape]


class Spotdetector(AbstractObject):  # Detectors/Spotdetector.jl:21-45
    kind = O_SPOT

    def __init__(self, width):
        shape = sh.QuadraticFlatMesh(width)
        sh.zrotate3d(shape, math.pi)
        super().__init__(shape)
        self.hw = width / 2
        self.data = np.zeros((0, 2))

    def empty(self):
        self.data = np.zeros((0, 2))


class PSFDetector(AbstractObject):  # Detectors/PSFDetector.jl:44-68
    kind = O_PSF

    def __init__(self, width):
        shape = sh.QuadraticFlatMesh(width)
        sh.zrotate3d(shape, math.pi)
        super().__init__(shape)
        self.data = np.zeros((0, 9))

    def empty(self):
        self.data = np.zeros((0, 9))

    # ---- read-out (PSFDetector.jl:91-237); the n x n coherent sum runs on the GPU engine (bmo_psf_intensity)
    def calc_local_pos(self):  # PSFDetector.jl:91-101
        loc = self.data[:, 0:3] - self.position()[None, :]
        o = self.orientation()
        return np.stack([loc @ o[:, 0], loc @ o[:, 2]], axis=1)

    def calc_local_lims(self, crop_factor=1.0, center="centroid"):  # PSFDetector.jl:116-144
        hits = self.calc_local_pos()
        xs, zs = hits[:, 0], hits[:, 1]
        if center == "centroid":
            w = self.data[:, 7]
            w_sum = w.sum()
            x0, z0 = (w * xs).sum() / w_sum, (w * zs).sum() / w_sum
        else:
            x0, z0 = (xs.min() + xs.max()) / 2, (zs.min() + zs.max()) / 2
        hwx, hwy = np.abs(xs - x0).max() * crop_factor, np.abs(zs - z0).max() * crop_factor
        return x0 - hwx, x0 + hwx, z0 - hwy, z0 + hwy

    def sample_axes(self, n=100, crop_factor=1.0, center="centroid", x_min=math.inf, x_max=math.inf, z_min=math.inf, z_max=math.inf,
                    x0_shift=0.0, z0_shift=0.0):
        """The (xs, zs) sample coordinates of intensity(psf; ...) PSFDetector.jl:205-217."""
        _x_min, _x_max, _z_min, _z_max = self.calc_local_lims(crop_factor=crop_factor, center=center)
        if x_min != math.inf and x_max != math.inf:
            _x_min, _x_max = x_min, x_max
        if z_min != math.inf and z_max != math.inf:
            _z_min, _z_max = z_min, z_max
        return la.linrange(_x_min, _x_max, n) + x0_shift, la.linrange(_z_min, _z_max, n) + z0_shift

    def intensity(self, n=100, device=0, _intensity_fn=None, **kw):
        """intensity(psf; n, crop_factor, center, x_min, ...) -> (xs, zs, I) with I[i, j] (PSFDetector.jl:190-237)."""
        from . import abi

        xs, zs = self.sample_axes(n=n, **kw)
        o = self.orientation()
        fn = _intensity_fn or (lambda *a: abi.psf_intensity(*a, device=device)[0])
        return xs, zs, fn(self.data, self.position(), o[:, 0], o[:, 2], xs, zs)


class Photodetector(AbstractObject):  # Detectors/Photodetector.jl:31-55
    """Flat square detector whose complex field `field[i, j]` (local x = pd.x[i], local y = pd.y[j]) is the superposition of all
    GaussianBeamlets that hit it (Photodetector.jl:69-107).  The trace records the hits; solve_system adds their field with the
    engine's bmo_photodetector_field right after the solve, so `field` reads like the reference's after solve_system!."""
    kind = O_PHOTODETECTOR

    def __init__(self, width, n):
        shape = sh.QuadraticFlatMesh(width)
        super().__init__(shape)
        sz = float(np.max(shape.vertices))
        self.x = la.linrange(-sz, sz, n)
        self.y = la.linrange(-sz, sz, n)
        self.field = np.zeros((n, n), dtype=np.complex128)

    def empty(self):  # empty!(pd) Photodetector.jl:119
        self.field[...] = 0

    def resolution(self, n):  # photodetector_resolution! Photodetector.jl:126-131
        self.x = la.linrange(self.x[0], self.x[-1], n)
        self.y = la.linrange(self.y[0], self.y[-1], n)
        self.field = np.zeros((n, n), dtype=np.complex128)

    def intensity(self):  # intensity.(pd.field), OpticUtils.jl:108
        return (self.field.real ** 2 + self.field.imag ** 2) / (2 * la.Z_vacuum)

    def optical_power(self):  # trapz((pd.x, pd.y), intensity(pd)) Photodetector.jl:116 (Trapz.jl: trapezoid rule along x, then y)
        I = self.intensity()
        inner = np.sum((I[1:, :] + I[:-1, :]) * np.diff(self.x)[:, None], axis=0) / 2
        return float(np.sum((inner[1:] + inner[:-1]) * np.diff(self.y)) / 2)


class IntersectableObject(AbstractObject):  # Intersectable.jl:10-15
    kind = O_INTERSECTABLE


class NonInteractableObject(AbstractObject):  # NonInteractable.jl:14-22
    kind = O_NONINTERACTABLE


class PolarizationFilter(AbstractObject):  # Polarizers/PolarizationFilter.jl:6-28
    kind = O_POLARIZER

    def __init__(self, edge_length, cutoff_strength=np.finfo(np.float64).eps, jones=None):
        shape = sh.QuadraticFlatMesh(edge_length)
        sh.zrotate3d(shape, math.pi)
        shape.set_new_origin3d()
        super().__init__(shape)
        # XZBasis(1, 0, 0, 0) PolarizedRays.jl:151: [j11 0 j12; 0 1 0; j21 0 j22]
        self.jones = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.complex128) if jones is None else np.asarray(jones, dtype=np.complex128)
        self.cutoff = float(cutoff_strength)
