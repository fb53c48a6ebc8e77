"""Collect gpurun_out/prof_round/ (tools/profile_round.sh) into profiles/<tag>_*: the step kernels' rows of the rocprofv3 kernel stats, the
bench line measured under the profiler, the counter summaries, and <tag>_traffic.json (what bench.py's `traffic` fields read).
    python tools/collect_profiles.py r03"""
import ast, csv, glob, json, os, re, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_round")
dst = os.path.join(ROOT, "profiles")
traffic = {}
for wdir in sorted(glob.glob(src + "/*/")):
    w = os.path.basename(wdir.rstrip("/"))
    ks = os.path.join(wdir, "kernel_stats.csv")
    if os.path.exists(ks):
        rows = list(csv.reader(open(ks)))
        keep = [rows[0]] + [r for r in rows[1:] if float(r[4]) >= 0.5 or "step_kernel" in r[0]]
        with open(os.path.join(dst, f"{tag}_kernel_stats_{w}.csv"), "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_ALL).writerows([[re.sub(r"\(anonymous namespace\)::", "", c)[:160] for c in r] for r in keep])
    bj = os.path.join(wdir, "bench_under_rocprof.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        d = json.loads(open(bj).read().strip().splitlines()[-1])
        json.dump(d, open(os.path.join(dst, f"{tag}_bench_under_rocprof_{w}.json"), "w"), indent=1)
    tot = {}
    for name in ("fetch", "write"):
        p = os.path.join(wdir, name + "_summary.txt")
        if not os.path.exists(p):
            continue
        m = re.search(r"total ms [\d.]+ (\{.*\})", open(p).read())
        if m:
            tot.update({k: float(v) for k, v in ast.literal_eval(m.group(1)).items()})
        launches = sum(1 for ln in open(p) if re.match(r"^\d+ \d+ ", ln))
        assert tot.get("launches", launches) == launches, f"{w}: the FETCH and WRITE passes saw different launch counts ({tot['launches']} vs {launches})"
        tot["launches"] = launches
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB (1024 B); bench.py doubles FETCH_SIZE (MI355X_MICROARCH.md "HBM": gfx950 counts
        # the 128-byte requests of a coalesced read at 64 B)
        traffic[w] = {"fetch_size_bytes": tot["FETCH_SIZE"] * 1024, "write_size_bytes": tot["WRITE_SIZE"] * 1024, "launches": tot["launches"],
                      "source": f"profiles/{tag}_pmc_{w}.txt: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over the step kernels of ONE solve, python3 bench.py --workload {w} --steps 1 --warmup 0"}
    # the VALU side of the accounting (bench.py's roofline.valu_issue): vector instructions per solve and the FP64 arithmetic among them
    p3 = os.path.join(wdir, "sq3_summary.txt")
    if w in traffic and os.path.exists(p3):
        m = re.search(r"total ms [\d.]+ (\{.*\})", open(p3).read())
        if m:
            c = {k: float(v) for k, v in ast.literal_eval(m.group(1)).items()}
            if c.get("SQ_INSTS_VALU"):
                traffic[w]["valu_wave_instructions"] = c["SQ_INSTS_VALU"]
                traffic[w]["valu_fp64_arithmetic"] = sum(c.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
                traffic[w]["valu_source"] = f"profiles/{tag}_pmc_{w}.txt, pass sq3 (SQ_INSTS_VALU, SQ_INSTS_VALU_{{ADD,MUL,FMA,TRANS}}_F64 over the step kernels of ONE solve)"
    with open(os.path.join(dst, f"{tag}_pmc_{w}.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc passes of ONE solve of workload {w} (tools/profile_round.sh; per launch of the step kernels, then totals).\n# SQ cycle counters are in quad-cycles; FETCH_SIZE / WRITE_SIZE in KB.\n")
        for name in ("sq", "sq2", "sq3", "fetch", "write"):
            p = os.path.join(wdir, name + "_summary.txt")
            if os.path.exists(p):
                f.write(f"## pass {name}\n" + open(p).read())
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print("collected", sorted(traffic))
