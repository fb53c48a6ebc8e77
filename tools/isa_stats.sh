#!/bin/bash
# usage: tools/isa_stats.sh <kernel-mangled-substring> [extra hipcc flags...] ; prints resource usage + load mix of one kernel
K=$1; shift
mkdir -p /tmp/st && cd /root/repo/beamletoptics.jl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills -Wno-unused-result "$@" -I ../../include -o /tmp/st/lib.so bmo_engine.hip -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|$K" -A8 | grep -E "error|VGPRs:|SGPRs:|Scratch|Occupancy" | head -8
cd /tmp/st; L=$(grep -n "^_ZN12_GLOBAL__N_1$K.*:" *gfx950.s | head -1 | cut -d: -f1); awk -v L=$L 'NR>=L && !done {print} /s_endpgm/ && NR>=L {done=1}' *gfx950.s > /tmp/k.s; wc -l /tmp/k.s
for p in s_load_dword global_load flat_load ds_read v_readfirstlane s_waitcnt s_cbranch; do echo $p $(grep -c "$p" /tmp/k.s); done
