"""Builds profiles/<tag>_hbm_traffic_and_sq_pmc.json and copies the kernel-stats CSV from the rocprofv3 output of collect.sh.
usage: python profiles/tools/summarize.py r1_e"""
import collections, csv, glob, json, shutil, sys
tag = sys.argv[1]
out = {"note": "rocprofv3 --pmc passes on MI355X, bench.py --steps 2 (BASELINE config[1] synthetic: 19,686 windows); FETCH_SIZE/WRITE_SIZE in KB per launch "
               "(averages). On gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): fetch_bytes_corrected doubles it. "
               "SQ counters are sums over one launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles.", "kernels": {}}
import re
def short(name):   # "void mirp::fold_lds_kernel<0>(FoldParams const*, ...)" -> "mirp::fold_lds_kernel"
    return re.sub(r"<.*>$", "", name.split("(")[0].replace("void ", "").strip())
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, ctr.split("_")[0].lower()))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr: acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        e = out["kernels"].setdefault(k, {})
        e[ctr + "_KB_avg"] = sum(v) / len(v); e[ctr + "_launches"] = len(v)
for k, e in out["kernels"].items():
    e["fetch_bytes_corrected"] = 2 * 1024 * e.get("FETCH_SIZE_KB_avg", 0.0)
    e["write_bytes"] = 1024 * e.get("WRITE_SIZE_KB_avg", 0.0)
for sub in ("sq", "sq2"):
    fs = glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, sub))
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[k].add(r["Dispatch_Id"])
    for k in ("mirp::fold_lds_kernel", "mirp::fold_lds_epilogue_kernel"):
        if k in acc:
            out.setdefault(k.split("::")[1] + "_sq_per_launch", {}).update({c: v / len(nl[k]) for c, v in acc[k].items()})
json.dump(out, open("profiles/%s_hbm_traffic_and_sq_pmc.json" % tag, "w"), indent=1, sort_keys=True)
shutil.copy(glob.glob("gpurun_out/%s_stats/*/*kernel_stats.csv" % tag)[0], "profiles/%s_kernel_stats.csv" % tag)
f185 = glob.glob("gpurun_out/%s_stats185/*/*kernel_stats.csv" % tag)
if f185:
    shutil.copy(f185[0], "profiles/%s_vienna185_kernel_stats.csv" % tag)
print(json.dumps({k: v for k, v in out.items() if k.endswith("per_launch")}, indent=1))
print({k: v for k, v in out["kernels"].items() if "fold" in k})
