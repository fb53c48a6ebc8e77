#!/bin/bash
# tools/ab_rmax.sh <rays> <r_max> lib... : kernel time of C2 with a given r_max for several engine builds
rays=$1; rmax=$2; shift 2
for lib in "$@"; do BMO_ENGINE_LIB=$PWD/$lib python - <<PY
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bmo_amd as bmo, scenes
system = scenes.c2_scene()[0]; b = scenes.c2_bundle($rays)
sc = bmo.CompiledScene(system, b.lambdas); eng = bmo.Engine(sc, 0); dev = eng.upload(b)
best = 1e9
for rep in range(4):
    res = eng.trace_device(dev, $rmax); k, t, n = eng.result_timing(res); eng.free_result(res); best = min(best, k)
print("%-40s r_max $rmax kernels %.3f ms launches %d" % ("$lib", best, n))
PY
done
