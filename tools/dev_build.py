"""(CPU, cross-compile) Developer build of the Ray step kernels only (-DBMO_DEV_RAY_LDS_ONLY, ~1 min) with extra flags, for tools/ab.sh:
    python tools/dev_build.py <name> [-DFLAG ...]      ->  build_ab/libbmo_<name>.so, prints registers / scratch / occupancy of its step kernels"""
import os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import __graft_entry__ as g

name, extra = sys.argv[1], sys.argv[2:]
dev = [] if any(f.startswith("-DBMO_DEV_") for f in extra) else ["-DBMO_DEV_RAY_LDS_ONLY"]
os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
out = os.path.join(ROOT, "build_ab", f"libbmo_{name}.so")
cmd = [g.HIPCC] + g.HIP_FLAGS + dev + extra + ["-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(g.CSRC, "bmo_engine.hip"), "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run(cmd, capture_output=True, text=True)
cur, row = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur, row = m.group(1), {}
    m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur and "step_kernel" in cur:
        row[m.group(1).split()[0]] = int(m.group(2))
        if len(row) == 3:
            short = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
            print("%-14s %-46s VGPR %3d scratch %4d B occupancy %d" % (name, short[:46], row["VGPRs"], row["ScratchSize"], row["Occupancy"]))
    if " error" in line:
        print(line)
sys.exit(r.returncode)
