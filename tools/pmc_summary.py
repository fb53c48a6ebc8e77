"""Summarise a rocprofv3 --pmc counter_collection.csv per step_kernel launch."""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "step_kernel" not in r["Kernel_Name"]:
        continue
    k = int(r["Dispatch_Id"])
    per.setdefault(k, {"grid": int(r["Grid_Size"]), "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                       "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "scratch": r["Scratch_Size"], "lds": r["LDS_Block_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in per.values() for k in v if k.isupper()})
print("launch grid ms " + " ".join(names))
tot = collections.defaultdict(float)
for i, (k, v) in enumerate(per.items()):
    print(i, v["grid"], "%.3f" % v["t"], " ".join("%.3g" % v.get(n, 0) for n in names))
    for n in names:
        tot[n] += v.get(n, 0)
    tot["t"] += v["t"]
print("total ms %.3f" % tot["t"], {n: "%.4g" % tot[n] for n in names})
v0 = next(iter(per.values()))
print("regs", v0["vgpr"], v0["agpr"], "scratch", v0["scratch"], "lds", v0["lds"])
