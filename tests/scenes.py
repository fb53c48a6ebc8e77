"""Synthetic scenes and ray bundles of the BASELINE configs, built with the host mirror.

C1: flat mirror + 2 spherical-SDF lenses.            C2/C3: miniscope train (docs/src/assets/ms_assets/ucla.jl:27-100
of the reference) + ThinBeamsplitter + 2 Spotdetectors = 10 elements, mesh + SDF.   C4: 3 singlets (6 surfaces) + stop.
C5: C2's train repeated x3 with fold mirrors = 32 elements.
All bundles are deterministic (PCG64 seed 20251003).
"""
import math

import numpy as np

import bmo_amd as bmo

mm = 1e-3
SEED = 20251003
NBK7_TABLE = ([532e-9, 1064e-9], [1.5195, 1.5066])  # ucla.jl:5


def nbk7():
    return bmo.DiscreteRefractiveIndex(*NBK7_TABLE)


def disc_bundle(n, center, direction, diameter, lam=1.064e-6, jitter=2e-3, e1=None, seed=SEED, cone=None):
    """Fibonacci disc (BeamGroups.jl:232-243) + per-ray angular jitter (uniform in [0, jitter] rad, or a filled cone)."""
    center = np.asarray(center, dtype=np.float64)
    d = bmo.linalg.normalize(np.asarray(direction, dtype=np.float64))
    e1 = bmo.linalg.perpendicular(d) if e1 is None else np.asarray(e1, dtype=np.float64)
    e2 = bmo.linalg.normalize(np.cross(d, e1))
    k = np.arange(n)
    rho = np.sqrt((k + 0.5) / n)
    phi = k * (2 * math.pi / (1 + math.sqrt(5)))
    r = diameter / 2 * rho
    pos = center[None, :] + (r * np.cos(phi))[:, None] * e1[None, :] + (r * np.sin(phi))[:, None] * e2[None, :]
    rng = np.random.Generator(np.random.PCG64(seed))
    if cone is not None:
        th = cone * np.sqrt(rng.random(n))
    else:
        th = jitter * rng.random(n)
    az = 2 * math.pi * rng.random(n)
    dirs = (np.cos(th))[:, None] * d[None, :] + (np.sin(th) * np.cos(az))[:, None] * e1[None, :] + (np.sin(th) * np.sin(az))[:, None] * e2[None, :]
    return bmo.RayBundle.rays(pos, dirs, lam)


# ------------------------------------------------------------------------------------ C1
def c1_scene():
    """SquarePlanoMirror2D(50 mm) folded 45 deg + two bi-convex SphericalLens(r, -r, l, 25.4 mm, n = 1.5)."""
    l1 = bmo.SphericalLens(60 * mm, -60 * mm, 6 * mm, 25.4 * mm, 1.5)
    l2 = bmo.SphericalLens(80 * mm, -80 * mm, 5 * mm, 25.4 * mm, 1.5)
    bmo.translate3d(l2, [0, 30 * mm, 0])
    mirror = bmo.SquarePlanoMirror2D(50 * mm)
    bmo.xrotate3d(mirror, math.radians(45))
    bmo.translate3d(mirror, [0, 70 * mm, 0])
    system = bmo.System([mirror, l1, l2])
    return system, dict(mirror=mirror, l1=l1, l2=l2)


def c1_bundle(n=1000):
    return disc_bundle(n, center=[0, -0.05, 0], direction=[0, 1, 0], diameter=0.8 * 25.4 * mm, lam=1.064e-6, e1=[1, 0, 0])


# ------------------------------------------------------------------------------------ C2
def miniscope_objects():
    """The lens train of ucla.jl:27-100 (without the STL housing): 7 leaf objects."""
    NBK7 = nbk7()
    obj_lens_1 = bmo.Lens(bmo.CircularFlatSurface(2 * 1.144 * mm), bmo.SphericalSurface(-1.448 * mm, 2 * 1.144 * mm), 1.3 * mm, NBK7)
    s1 = bmo.SphericalSurface(38.184 * mm, 2 * 1.840 * mm, 2 * 2.380 * mm)
    s2 = bmo.SphericalSurface(3.467 * mm, 2 * 2.060 * mm, 2 * 2.380 * mm)
    s3 = bmo.SphericalSurface(-5.020 * mm, 2 * 2.380 * mm)
    dl11 = bmo.Lens(s1, s2, 0.5 * mm, NBK7)
    dl12 = bmo.Lens(s2, s3, 2.5 * mm, NBK7)
    bmo.translate3d(dl12, [0, dl11.thickness, 0])
    obj_lens_2 = bmo.DoubletLens(dl11, dl12)
    s1 = bmo.SphericalSurface(7.744 * mm, 2 * 2.812 * mm, 2 * 3 * mm)
    s2 = bmo.SphericalSurface(-3.642 * mm, 2 * 3 * mm)
    s3 = bmo.SphericalSurface(-14.413 * mm, 2 * 2.812 * mm, 2 * 3 * mm)
    dl21 = bmo.Lens(s1, s2, 3.4 * mm, NBK7)
    dl22 = bmo.Lens(s2, s3, 1.0 * mm, NBK7)
    bmo.translate3d(dl22, [0, dl21.thickness, 0])
    tube_lens = bmo.DoubletLens(dl21, dl22)
    bmo.translate_to3d(obj_lens_2, [0, bmo.position(obj_lens_1)[1] + obj_lens_1.thickness + 3.344 * mm, 0])
    bmo.translate_to3d(tube_lens, [0, bmo.position(obj_lens_2)[1] + obj_lens_2.thickness + 2 * mm, 0])
    objective_group = bmo.ObjectGroup([obj_lens_1, obj_lens_2, tube_lens])
    bmo.xrotate3d(objective_group, math.radians(90))
    shape = bmo.CuboidMesh(8 * mm, 1 * mm, 8.5 * mm)
    bmo.translate3d(shape, [-4 * mm, 0.0, -4.25 * mm])
    shape.set_new_origin3d()
    bmo.translate3d(shape, [0, 0, 18.677 * mm])
    bmo.xrotate3d(shape, math.radians(45))
    filt = bmo.Lens(shape, NBK7)
    ef_1 = bmo.Lens(bmo.PlanoSurfaceSDF(1 * mm, 4 * mm), NBK7)
    ef_2 = bmo.Lens(bmo.PlanoSurfaceSDF(1 * mm, 4 * mm), NBK7)
    collect_lens = bmo.Lens(bmo.SphericalSurface(6.580 * mm, 4.5 * mm), bmo.SphericalSurface(-6.580 * mm, 4.5 * mm), 2.6 * mm, NBK7)
    bmo.translate3d(collect_lens, [0, bmo.position(ef_1)[1] + ef_1.thickness + 0.1 * mm, 0])
    bmo.translate3d(ef_2, [0, bmo.position(collect_lens)[1] + collect_lens.thickness + 0.25 * mm, 0])
    collect_group = bmo.ObjectGroup([ef_1, collect_lens, ef_2])
    bmo.xrotate3d(collect_group, math.radians(90))
    bmo.translate3d(collect_group, [0, 0.332 * mm, 21.937 * mm])
    return [objective_group, filt, collect_group]


def c2_scene():
    """Miniscope train + ThinBeamsplitter(10 mm) at 45 deg + one Spotdetector(5 mm) per arm = 10 leaf objects."""
    objs = miniscope_objects()
    y0 = 0.332 * mm
    bs = bmo.ThinBeamsplitter(10 * mm)
    bmo.xrotate3d(bs, math.radians(45))
    bmo.translate3d(bs, [0, y0, 30 * mm])
    det_t = bmo.Spotdetector(5 * mm)
    bmo.xrotate3d(det_t, math.radians(90))
    bmo.translate3d(det_t, [0, y0, 34 * mm])
    det_r = bmo.Spotdetector(5 * mm)
    bmo.translate3d(det_r, [0, y0 - 4 * mm, 30 * mm])
    system = bmo.System(objs + [bs, det_t, det_r])
    return system, dict(bs=bs, det_t=det_t, det_r=det_r)


def c2_bundle(n, lam=1.064e-6, seed=SEED):
    """Extended object 0.3 mm across at the miniscope's object plane, rays filling a 0.25 rad cone (NA ~ 0.25)."""
    return disc_bundle(n, center=[0, 0, -0.77 * mm], direction=[0, 0, 1], diameter=0.3 * mm, lam=lam, e1=[1, 0, 0], cone=0.25, seed=seed)


def c2_survey_bundle(n, lam=1.064e-6, seed=SEED):
    """SURVEY.md §8(d)'s literal bundle for config 2: the Fibonacci disc of UniformDiscSource (BeamGroups.jl:232-243) with diameter
    0.8 x the first clear aperture (2 x 1.144 mm), directions along the optical axis plus a per-ray angular jitter of at most 2 mrad."""
    return disc_bundle(n, center=[0, 0, -0.77 * mm], direction=[0, 0, 1], diameter=0.8 * 2.288 * mm, lam=lam, e1=[1, 0, 0], jitter=2e-3, seed=seed)


def c2_vignetted_bundle(n, lam=1.064e-6, seed=SEED, diameter=2.0 * mm, cone=0.6):
    """The ragged counterpart of c2_bundle: an object disc and a cone wider than the miniscope accepts, so part
    of the bundle misses the first aperture, is clipped at lens rims / mechanical rings or leaves the train between elements —
    beams end after different bounce counts and not every root reaches the splitter."""
    return disc_bundle(n, center=[0, 0, -0.77 * mm], direction=[0, 0, 1], diameter=diameter, lam=lam, e1=[1, 0, 0], cone=cone, seed=seed)


# ------------------------------------------------------------------------------------ C4
def c4_scene():
    """Three singlets (bi-convex, meniscus, plano-convex => 6 refracting surfaces) + IntersectableObject end stop."""
    l1 = bmo.SphericalLens(60 * mm, -60 * mm, 6 * mm, 25.4 * mm, 1.5)
    l2 = bmo.SphericalLens(30 * mm, 60 * mm, 2.5 * mm, 25.4 * mm, 1.6)  # l0 <= 0 with same-sign radii => MeniscusLensSDF
    assert isinstance(l2.shape, bmo.MeniscusLensSDF)
    l3 = bmo.SphericalLens(math.inf, -40 * mm, 5 * mm, 25.4 * mm, 1.7)
    bmo.translate3d(l2, [0, 15 * mm, 0])
    bmo.translate3d(l3, [0, 30 * mm, 0])
    stop = bmo.IntersectableObject(bmo.QuadraticFlatMesh(40 * mm))
    bmo.translate3d(stop, [0, 60 * mm, 0])
    for o in (l1, l2, l3):  # a small tilt so s/p bases are not degenerate everywhere
        bmo.xrotate3d(o, math.radians(3))
    return bmo.System([l1, l2, l3, stop]), dict(l1=l1, l2=l2, l3=l3, stop=stop)


def polarized_bundle(n, center, direction, diameter, lam=1.064e-6, jitter=2e-3, e1=None, seed=SEED):
    """disc_bundle + E0 = unit linear polarization at 45 deg in the (e1, e2) plane, made orthogonal to each ray's dir."""
    b = disc_bundle(n, center, direction, diameter, lam, jitter, e1, seed)
    d = bmo.linalg.normalize(np.asarray(direction, dtype=np.float64))
    e1 = bmo.linalg.perpendicular(d) if e1 is None else np.asarray(e1, dtype=np.float64)
    e2 = bmo.linalg.normalize(np.cross(d, e1))
    dirs = b.planes[3:6].T
    E = (e1 + e2)[None, :] / math.sqrt(2) * np.ones((n, 1))
    for _ in range(3):  # Gram-Schmidt to 1e-16
        E = E - (E * dirs).sum(axis=1)[:, None] * dirs
    E = E / np.linalg.norm(E, axis=1)[:, None]
    P = np.zeros((14, n))
    P[:8] = b.planes
    P[8:14:2] = E.T
    return bmo.RayBundle(bmo.BEAM_POLARIZED, P)


def c4_bundle(n):
    return polarized_bundle(n, center=[0, -0.03, 0], direction=[0, 1, 0], diameter=0.8 * 25.4 * mm, e1=[1, 0, 0])


# ------------------------------------------------------------------------------------ C3
def gaussian_bundle(n, center, direction, diameter, lam=1.064e-6, w0=50e-6, support=(1.0, 0.0, 0.0), M2=1.0, P0=1e-3, z0=0.0, cone=None,
                    jitter=2e-3, seed=SEED):
    """N GaussianBeamlet(pos, dir, lam, w0; support) built like the reference constructor (Gaussian.jl:215-256), vectorised."""
    b = disc_bundle(n, center, direction, diameter, lam, jitter, e1=support, seed=seed, cone=cone)
    pos, d = b.planes[0:3].T, b.planes[3:6].T
    s1 = bmo.linalg.normalize(np.asarray(support, dtype=np.float64))
    tan_t = math.tan(M2 * lam / (math.pi * w0))
    dd = d + s1[None, :] * tan_t
    dd = dd / np.sqrt((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2])[:, None]
    P = np.zeros((25, n))
    P[0:3], P[3:6] = pos.T, d.T
    P[6:9], P[9:12] = (pos + s1[None, :] * w0).T, d.T
    P[12:15], P[15:18] = (pos + s1[None, :] * (-z0 * tan_t)).T, dd.T
    P[18], P[19], P[20] = lam, 1.0, w0
    P[21] = math.sqrt(2 * (2 * P0 / (math.pi * w0 ** 2)) * bmo.linalg.Z_vacuum)
    return bmo.RayBundle(bmo.BEAM_GAUSSIAN, P)


# ------------------------------------------------------------------------------------ C5
C5_X = (-12 * mm, 0.0, 12 * mm)


def c5_scene():
    """32 leaf objects: the miniscope train x3 (side by side in x), per train a 45 deg RoundPlanoMirror fold and a
    RightAnglePrismMirror fold, one ThinBeamsplitter in the middle arm, two Spotdetectors, two IntersectableObject baffles."""
    objs = []
    y0 = 0.332 * mm
    for x0 in C5_X:
        grp = bmo.ObjectGroup(miniscope_objects())
        bmo.translate3d(grp, [x0, 0, 0])
        objs.append(grp)
    for x0 in C5_X:
        fold = bmo.RoundPlanoMirror(8 * mm, 2 * mm)
        bmo.xrotate3d(fold, math.radians(45))
        bmo.translate3d(fold, [x0, y0, 30 * mm])
        objs.append(fold)
    for x0 in C5_X:
        prism = bmo.RightAnglePrismMirror(8 * mm, 8 * mm)
        bmo.xrotate3d(prism, math.radians(225))
        bmo.translate3d(prism, [x0, y0 - 10 * mm, 30 * mm])
        objs.append(prism)
    bs = bmo.ThinBeamsplitter(6 * mm)
    bmo.xrotate3d(bs, math.radians(45))
    bmo.translate3d(bs, [0, y0 - 10 * mm, 38 * mm])
    det_t = bmo.Spotdetector(40 * mm)
    bmo.xrotate3d(det_t, math.radians(90))
    bmo.translate3d(det_t, [0, y0 - 10 * mm, 45 * mm])
    det_r = bmo.Spotdetector(10 * mm)
    bmo.translate3d(det_r, [0, y0 - 16 * mm, 38 * mm])
    baffles = []
    for x0 in (-6 * mm, 6 * mm):
        b = bmo.IntersectableObject(bmo.RectangularFlatMesh(30 * mm, 30 * mm))
        bmo.zrotate3d(b, math.radians(90))
        bmo.translate3d(b, [x0, 0, 12 * mm])
        baffles.append(b)
    system = bmo.System(objs + [bs, det_t, det_r] + baffles)
    return system, dict(bs=bs, det_t=det_t, det_r=det_r)


def c5_bundle(n, lam=1.064e-6, seed=SEED):
    """Ray k feeds train k mod 3 (object plane of that train)."""
    b = c2_bundle(n, lam, seed=seed)
    P = b.planes.copy()
    P[0] += np.array(C5_X)[np.arange(n) % 3]
    return bmo.RayBundle(b.kind, P)


def c3_bundle(n):
    return gaussian_bundle(n, center=[0, 0, -0.77 * mm], direction=[0, 0, 1], diameter=0.3 * mm, w0=50e-6, support=(1.0, 0.0, 0.0), cone=0.1)
