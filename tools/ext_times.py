"""Kernel / solve time of the step-kernel variants the bench workloads do not reach (library named by BMO_ENGINE_LIB, default the in-tree build):
the extended-shape levels (1: meniscus lenses, 2: aspheres) under Rays, PolarizedRays and GaussianBeamlets, fresh solves and retraces.
    python tools/ext_times.py [rays [substring of the case names to run]]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bmo_amd as bmo
import scenes
from test_asphere_system import build as asphere_lenses

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
mm = 1e-3


def asphere_system():
    det = bmo.Spotdetector(10 * mm)
    bmo.translate3d(det, [0, 4.2 * mm, 0])
    return bmo.System(list(asphere_lenses()) + [det])


def singlets():  # C4's train with a bi-convex lens in the meniscus' place: the plain-shapes level under PolarizedRays
    import math
    l1 = bmo.SphericalLens(60 * mm, -60 * mm, 6 * mm, 25.4 * mm, 1.5)
    l2 = bmo.SphericalLens(80 * mm, -60 * mm, 2.5 * mm, 25.4 * mm, 1.6)
    l3 = bmo.SphericalLens(math.inf, -40 * mm, 5 * mm, 25.4 * mm, 1.7)
    bmo.translate3d(l2, [0, 15 * mm, 0])
    bmo.translate3d(l3, [0, 30 * mm, 0])
    stop = bmo.IntersectableObject(bmo.QuadraticFlatMesh(40 * mm))
    bmo.translate3d(stop, [0, 60 * mm, 0])
    for o in (l1, l2, l3):
        bmo.xrotate3d(o, math.radians(3))
    return bmo.System([l1, l2, l3, stop])


c4 = scenes.c4_scene()[0]
cases = [
    ("ray  ext0 (C2 scene, 8(d) bundle)", scenes.c2_scene()[0], scenes.c2_survey_bundle(n)),
    ("pol  ext0 (three singlets)", singlets(), scenes.c4_bundle(n)),
    ("gauss ext0 (C2 scene)", scenes.c2_scene()[0], scenes.c3_bundle(n // 4)),
    ("ray  ext1 (C4 scene)", c4, scenes.disc_bundle(n, center=[0, -20 * mm, 0], direction=[0, 1, 0], diameter=8 * mm)),
    ("pol  ext1 (C4)", c4, scenes.c4_bundle(n)),
    ("gauss ext1 (C4 scene)", c4, scenes.gaussian_bundle(n // 4, center=[0, -20 * mm, 0], direction=[0, 1, 0], diameter=6 * mm)),
    ("ray  ext2 (asphere objective)", asphere_system(), scenes.disc_bundle(n, center=[0, -0.5 * mm, 0], direction=[0, 1, 0], diameter=1.3 * mm, lam=0.5876e-6, jitter=0.05)),
    ("pol  ext2 (asphere objective)", asphere_system(), scenes.polarized_bundle(n, center=[0, -0.5 * mm, 0], direction=[0, 1, 0], diameter=1.3 * mm, lam=0.5876e-6, jitter=0.05)),
    ("gauss ext2 (asphere objective)", asphere_system(), scenes.gaussian_bundle(n // 4, center=[0, -0.5 * mm, 0], direction=[0, 1, 0], diameter=1.0 * mm, lam=0.5876e-6, w0=20e-6)),
]
lib = os.environ.get("BMO_ENGINE_LIB", "default")[-24:]
only = sys.argv[2] if len(sys.argv) > 2 else ""
for name, system, bundle in cases:
    if only not in name:
        continue
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    dev = eng.upload(bundle)
    best, res = (1e9, 0, 0), None
    for rep in range(4):
        if res is not None:
            eng.free_result(res)
        res = eng.trace_device(dev, 50)
        kms, tms, nl = eng.result_timing(res)
        if kms < best[0]:
            best = (kms, tms, nl)
    rbest = (1e9, 0, 0)
    for rep in range(3):  # retrace of the solved beams (nothing moved: every stored intersection is confirmed)
        r2 = eng.retrace_device(dev, res, 50)
        kms, tms, nl = eng.result_timing(r2)
        eng.free_result(r2)
        if kms < rbest[0]:
            rbest = (kms, tms, nl)
    eng.free_result(res)
    eng.free_batch(dev)
    print("%-24s %-32s beams %7d  fresh: kernels %7.3f ms total %7.3f ms (%d launches)   retrace: kernels %7.3f ms total %7.3f ms" % (lib, name, bundle.n, *best, rbest[0], rbest[1]), flush=True)
    eng.close()
