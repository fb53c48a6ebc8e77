"""Aspheric and cylindrical lens known-answer tests of the reference (runtests.jl:1522-1579, 1697-1740) on the oracle,
plus lane-code == oracle parity on scenes that contain those shapes."""
import math

import numpy as np

import bmo_amd as bmo
from parity import compare, emu_trace
from scenes import disc_bundle, mm

R, K, D, CT, N = 50.3583e-3, -0.789119, 50e-3, 10.2e-3, 1.5036
A = [0, 2.10405e-7 * (1e3) ** 3, 1.76468e-11 * (1e3) ** 5, 1.02641e-15 * (1e3) ** 7]  # Thorlabs AL50100J


def al50100j():
    return bmo.Lens(bmo.EvenAsphericalSurface(R, D, K, A), CT, lambda x: N)


def working_distance(oracle, lens, offset_z, start_y=-1.0, r_max=100):
    beam = bmo.Beam(bmo.Ray([0.0, start_y, offset_z], [0.0, 1.0, 0]))
    oracle.solve_system(bmo.System([lens]), beam, r_max=r_max)
    r = beam.rays[-1]
    dist = -r.pos[2] / r.dir[2]
    return math.cos(math.asin(r.dir[2])) * dist


def test_asphere_surface_and_working_distance(oracle):  # runtests.jl:1529-1579
    lens = al50100j()
    beams = [bmo.Beam(bmo.Ray([0.0, -0.1, z], [0.0, 1.0, 0])) for z in np.linspace(-0.02, 0.02, 100)]
    oracle.solve_system(bmo.System([lens]), beams, r_max=40, threads=8)
    for b in beams:
        r0 = b.rays[0]
        hit_y = (r0.pos + r0.intersection.t * r0.dir)[1]
        assert abs(hit_y - bmo.shapes.aspheric_equation(r0.pos[2], 1 / R, K, A)) <= 1e-10
    assert abs(working_distance(oracle, lens, 0.02, start_y=-0.1, r_max=40) - 93.2e-3) <= 1e-4


def test_cylinder_lenses(oracle):  # runtests.jl:1711-1740
    r, d, h, ct = 5.2e-3, 10e-3, 20e-3, 5.9e-3
    lens = bmo.Lens(bmo.CylindricalSurface(r, d, h), ct, lambda n: 1.517)
    assert abs(lens.thickness - ct) <= 1.5e-8 * ct
    assert abs(bmo.thickness(lens.shape.sdfs[0]) - 2.12e-3) <= 1e-4
    assert abs(working_distance(oracle, lens, 0.05 * d / 2) - 6.1e-3) <= 1e-4
    lens = bmo.Lens(bmo.CylindricalSurface(-13.1e-3, 16e-3, 18e-3), 2.0e-3, lambda n: 1.517)
    assert abs(lens.thickness - 2.0e-3) <= 1.5e-8 * 2.0e-3


def asphere_cylinder_scene():
    asph = al50100j()
    cyl = bmo.Lens(bmo.CylindricalSurface(30e-3, 40e-3, 40e-3), 8e-3, lambda n: 1.517)
    bmo.translate3d(cyl, [0, 30 * mm, 0])
    ccyl = bmo.Lens(bmo.CylindricalSurface(-40e-3, 40e-3, 40e-3), 4e-3, lambda n: 1.6)
    bmo.translate3d(ccyl, [0, 55 * mm, 0])
    bmo.yrotate3d(ccyl, math.radians(20))
    det = bmo.Spotdetector(80 * mm)
    bmo.translate3d(det, [0, 120 * mm, 0])
    for o in (asph, cyl):
        bmo.xrotate3d(o, math.radians(2))
    return bmo.System([asph, cyl, ccyl, det])


def test_lane_code_matches_oracle_on_aspheres_and_cylinders(oracle):
    system = asphere_cylinder_scene()
    b = disc_bundle(400, center=[0, -0.05, 0], direction=[0, 1, 0], diameter=44 * mm, e1=[1, 0, 0], jitter=5e-3)
    sc = bmo.CompiledScene(system, b.lambdas)
    ref = oracle.trace(sc, b, 40, threads=8)
    compare(emu_trace(sc, b, 40), ref, 0.0, "asph+cyl")
    assert ref.det_count[0] > 300 and int(ref.node_nseg.max()) == 7
    off = bmo.CompiledScene(system, b.lambdas, cull=False)
    compare(emu_trace(off, b, 40), ref, 0.0, "asph+cyl cull-off")


AYL = [0, 1.1926075e-5 * (1e3) ** 3, -2.9323497e-9 * (1e3) ** 5, -1.8718889e-11 * (1e3) ** 7, -1.7009961e-14 * (1e3) ** 9,
       3.5481542e-17 * (1e3) ** 11, 6.5241296e-20 * (1e3) ** 13]  # Thorlabs AYL2520


def test_acylinder_lenses(oracle):  # runtests.jl:1744-1785
    lens = bmo.Lens(bmo.AcylindricalSurface(15.538e-3, 25e-3, 50e-3, -1.0, AYL), 7.5e-3, lambda n: 1.777)
    assert abs(lens.thickness - 7.5e-3) <= 1.5e-8 * 7.5e-3
    assert abs(working_distance(oracle, lens, 0.05 * 25e-3 / 2) - 15.8e-3) <= 1e-4
    lens = bmo.Lens(bmo.AcylindricalSurface(-15.538e-3, 25e-3, 50e-3, -1.0, AYL), 7.5e-3, lambda n: 1.777)
    assert abs(lens.thickness - 7.5e-3) <= 1.5e-8 * 7.5e-3


def test_lane_code_matches_oracle_on_acylinders(oracle):
    a1 = bmo.Lens(bmo.AcylindricalSurface(15.538e-3, 25e-3, 50e-3, -1.0, AYL), 7.5e-3, lambda n: 1.777)
    a2 = bmo.Lens(bmo.AcylindricalSurface(-15.538e-3, 25e-3, 50e-3, -1.0, AYL), 7.5e-3, lambda n: 1.6)
    bmo.translate3d(a2, [0, 20 * mm, 0])
    bmo.yrotate3d(a2, math.radians(35))
    bmo.xrotate3d(a1, math.radians(1.5))
    det = bmo.Spotdetector(80 * mm)
    bmo.translate3d(det, [0, 60 * mm, 0])
    system = bmo.System([a1, a2, det])
    b = disc_bundle(300, center=[0, -0.03, 0], direction=[0, 1, 0], diameter=22 * mm, e1=[1, 0, 0], jitter=5e-3)
    sc = bmo.CompiledScene(system, b.lambdas)
    ref = oracle.trace(sc, b, 40, threads=8)
    compare(emu_trace(sc, b, 40), ref, 0.0, "acyl")
    assert ref.det_count[0] > 200
