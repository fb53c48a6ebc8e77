"""SDF evaluations per record and bounce level on C2 (or its vignetted bundle: `emu_stats.py N c2v`, SURVEY's literal bundle: `emu_stats.py N c2s`), counted by the host emulator built
with -DBMO_EMU_STATS (CPU only)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bmo_amd as bmo, parity, scenes
from bmo_amd import abi
so = "/tmp/libbmo_emu_stats.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DBMO_EMU_STATS", "-shared", "-o", so, os.path.join(ROOT, "tests/emu/emu.cpp")])
emu = C.CDLL(so)
emu.bmo_emu_trace.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.RayBatch), C.POINTER(abi.TraceOpts), C.POINTER(C.c_void_p), C.POINTER(abi.ResultView)]
emu.bmo_emu_free.argtypes = [C.c_void_p]
parity._emu = emu
system, _ = scenes.c2_scene()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else "c2"
b = {"c2v": scenes.c2_vignetted_bundle, "c2s": scenes.c2_survey_bundle}.get(which, scenes.c2_bundle)(n)
parity.emu_trace(bmo.CompiledScene(system, b.lambdas), b, 20)
