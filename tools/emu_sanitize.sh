#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer build of the lane code's host emulator (tests/emu: csrc/bmo_lane.hpp compiled for the
# CPU) over the randomised parity corpus.  GPU sanitizers are not available on the pool; this is the CPU stand-in the advisor asked
# for after the round-1 `gauss_step` loop incident.  Restores the plain emulator build afterwards.
set -e
cd "$(dirname "$0")/.."
cp tests/emu/libbmo_emu.so /tmp/libbmo_emu_plain.so 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared \
    -o tests/emu/libbmo_emu.so tests/emu/emu.cpp
trap 'make -s -B -C tests/emu' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest tests/test_fuzz.py tests/test_leaf_kinds.py tests/test_cull.py tests/test_retrace.py \
    tests/test_retrace_stale.py tests/test_many_objects.py tests/test_degenerate_rays.py -x -q -m "not gpu" -p no:cacheprovider "$@"
