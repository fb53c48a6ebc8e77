"""Round 4's scheduling measures never touch a result (DESIGN.md §3 "Scheduling"): the order of the root rays in the first chunk
(`BMO_ROOT_ORDER`: the bundle's own order with candidate-set bins, the Morton order of the rays' chords through the scene's bounding sphere,
plain bundle order — read when a batch is uploaded), the reflected child a lane keeps for itself, and the order in which tiles are handed
out — from the end of the chunk in the first solve of a batch, slowest first by the previous solve's per-tile times from the second on.
Every combination gives the oracle's beams, rays, hits and detector rows, bit for bit; beam nodes keep the bundle's numbering whatever
the slot order is (`System.jl:463-468`: result order = bundle order x BFS order)."""
import os

import numpy as np
import pytest

import bmo_amd as bmo
import scenes
from parity import compare


def _ragged_case(n):
    """config 2's scene under the ragged bundle (directions drawn at random in a 0.6 rad cone: `auto` takes the Morton order for it)"""
    system, _ = scenes.c2_scene()
    b = scenes.c2_vignetted_bundle(n)
    return bmo.CompiledScene(system, b.lambdas), b


def _disc_case(n):
    """config 2's scene under SURVEY 8(d)'s collimated disc (a spiral numbering: `auto` keeps the bundle's own order)"""
    system, _ = scenes.c2_scene()
    b = scenes.c2_survey_bundle(n)
    return bmo.CompiledScene(system, b.lambdas), b


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ragged", "disc"])
def test_root_orders_and_tile_feedback_do_not_change_results(oracle, monkeypatch, case):
    scene, b = (_ragged_case if case == "ragged" else _disc_case)(8192)
    ref = oracle.trace(scene, b, 100, threads=16)
    eng = bmo.Engine(scene, 0)
    try:
        for order in ("auto", "chord", "mask", "none"):
            monkeypatch.setenv("BMO_ROOT_ORDER", order)
            dev = eng.upload(b)
            try:
                for solve in range(3):  # the second and third solve of a batch run their tiles in the order of the previous solve's times
                    res = eng.trace_device(dev, 100)
                    try:
                        compare(eng.result_view(res), ref, 0.0, f"{case}, root order {order}, solve {solve}")
                    finally:
                        eng.free_result(res)
            finally:
                eng.free_batch(dev)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ragged", "disc"])
def test_wide_kernels_give_the_same_results(oracle, monkeypatch, case):
    """The Ray kernels of the plain-shapes level exist twice — compiled for 3 and for 4 waves per SIMD (168 / 128 registers, `step_waves`) — and a
    launch picks by its size (`BMO_WIDE_MIN_WAVES`, default 4096 waves).  Same lane code, other register allocation and spills: same bits."""
    scene, b = (_ragged_case if case == "ragged" else _disc_case)(8192)
    ref = oracle.trace(scene, b, 100, threads=16)
    eng = bmo.Engine(scene, 0)
    try:
        dev = eng.upload(b)
        try:
            for wide_min in ("0", "1000000000", "0"):
                monkeypatch.setenv("BMO_WIDE_MIN_WAVES", wide_min)
                res = eng.trace_device(dev, 100)
                try:
                    compare(eng.result_view(res), ref, 0.0, f"{case}, BMO_WIDE_MIN_WAVES={wide_min}")
                finally:
                    eng.free_result(res)
        finally:
            eng.free_batch(dev)
    finally:
        eng.close()


@pytest.mark.gpu
def test_kept_reflected_children_through_a_splitter_chain(oracle):
    """Four thin splitters in a row (the transmitted beam meets the next one): a lane keeps ONE reflected child (StepParams::pend), a split while one waits goes to the next
    launch's chunk, kept children are taken up when their lane's beam ends — beam tree, order and detector rows as the oracle has them."""
    import math

    mm = 1e-3
    objs = []
    for q in range(4):
        bs = bmo.ThinBeamsplitter(30 * mm)
        bmo.xrotate3d(bs, math.radians(45 if q % 2 == 0 else -45))
        bmo.translate3d(bs, [0, (40 + 25 * q) * mm, 0])
        objs.append(bs)
    lens = bmo.SphericalLens(80 * mm, -80 * mm, 4 * mm, 25.4 * mm, 1.5)
    bmo.translate3d(lens, [0, 10 * mm, 0])
    det = bmo.Spotdetector(200 * mm)
    bmo.translate3d(det, [0, 160 * mm, 0])
    system = bmo.System([lens] + objs + [det])
    b = scenes.disc_bundle(8192, center=[0, -20 * mm, 0], direction=[0, 1, 0], diameter=8 * mm, lam=1.064e-6, cone=0.05)
    scene = bmo.CompiledScene(system, b.lambdas)
    ref = oracle.trace(scene, b, 100, threads=16)
    assert ref.n_nodes == 9 * b.n  # the transmitted beam meets the next splitter, the reflected one leaves: 1 + 2 x 4 beams per ray
    eng = bmo.Engine(scene, 0)
    try:
        dev = eng.upload(b)
        for solve in range(3):
            if solve == 2:  # ... and through the 4-waves-per-SIMD build of the kernel
                os.environ["BMO_WIDE_MIN_WAVES"] = "0"
            try:
                res = eng.trace_device(dev, 100)
            finally:
                os.environ.pop("BMO_WIDE_MIN_WAVES", None)
            compare(eng.result_view(res), ref, 0.0, f"splitter chain, solve {solve}")
            eng.free_result(res)
        eng.free_batch(dev)
    finally:
        eng.close()
