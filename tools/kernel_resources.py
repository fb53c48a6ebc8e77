"""Resource usage (VGPRs, SGPRs, scratch, occupancy) of the step kernels: python tools/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills", "-Wno-unused-result",
       "-I", os.path.join(ROOT, "include"), "-o", "/tmp/kr_lib.so", os.path.join(ROOT, "beamletoptics.jl_amd/csrc/bmo_engine.hip"),
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"\s(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if "step_kernel" in k or "psf" in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        print("%-58s VGPR %3d SGPR %3d scratch %4d occ %d" % (name[:58], v.get("VGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1)))
