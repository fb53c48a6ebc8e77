"""The C-ABI library loads and exports every symbol include/bmo.h declares (no compute calls: runs without a GPU),
and the product path fails loudly when the HIP extension is missing."""
import ctypes as C
import os
import re

import pytest

import bmo_amd as bmo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bmo.h")).read()
    return sorted(set(re.findall(r"^(?:int|double|const char\*)\s+(bmo_[a-z_]+)\s*\(", src, flags=re.M)))


def test_engine_exports_every_declared_symbol():
    lib = bmo.abi.load_engine()
    names = _declared()
    assert {"bmo_version", "bmo_scene_create", "bmo_trace", "bmo_trace_device", "bmo_result_view", "bmo_result_copy_hits"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.bmo_version() == bmo.abi.ABI_VERSION


def test_struct_sizes_match_header():
    # sizes the Julia / ctypes mirrors rely on
    assert C.sizeof(bmo.abi.Shape) == 288
    assert C.sizeof(bmo.abi.Object) == 200


def test_scene_create_validates_without_gpu():
    lib = bmo.abi.load_engine()
    from scenes import c1_scene

    system, _ = c1_scene()
    sc = bmo.CompiledScene(system, [1e-6])
    h = C.c_void_p()
    assert lib.bmo_scene_create(C.byref(sc.desc), C.byref(h)) == 0
    lib.bmo_scene_destroy(h)
    sc.desc.abi_version = 99
    assert lib.bmo_scene_create(C.byref(sc.desc), C.byref(h)) == -1  # BMO_ERR_INVALID
    assert b"abi" in lib.bmo_last_error()
    sc.desc.abi_version = bmo.abi.ABI_VERSION
    sc._objects[0].shape[0] = 10_000
    assert lib.bmo_scene_create(C.byref(sc.desc), C.byref(h)) == -1


def test_missing_engine_fails_loudly(monkeypatch):
    monkeypatch.setattr(bmo.abi, "_engine", None)
    monkeypatch.setattr(bmo.abi, "ENGINE_PATH", "/nonexistent/libbmo_hip.so")
    with pytest.raises(bmo.abi.EngineMissing):
        bmo.abi.load_engine()


def test_trace_without_gpu_is_an_error_not_a_fallback():
    lib = bmo.abi.load_engine()
    if lib.bmo_device_count() > 0:
        pytest.skip("a GPU is visible here")
    from scenes import c1_bundle, c1_scene

    system, _ = c1_scene()
    b = c1_bundle(4)
    sc = bmo.CompiledScene(system, b.lambdas)
    eng = bmo.Engine(sc, 0)
    with pytest.raises(RuntimeError, match="no HIP device"):
        eng.trace(b)
    eng.close()
