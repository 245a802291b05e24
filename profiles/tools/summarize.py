"""Builds profiles/<tag>_hbm_traffic_and_sq_pmc.json and copies the kernel-stats CSV from the rocprofv3 output of collect.sh.
usage: python profiles/tools/summarize.py r1_e"""
import collections, csv, glob, json, shutil, sys
tag = sys.argv[1]
out = {"note": "rocprofv3 --pmc passes on MI355X, bench.py --steps 2 (BASELINE config[1] synthetic: 19,686 windows); FETCH_SIZE/WRITE_SIZE in KB per launch "
               "(averages). On gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): fetch_bytes_corrected doubles it. "
               "SQ counters are sums over one launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles.", "kernels": {}}
import re
def short(name):   # "void mirp::fold_lds_kernel<0, true>(FoldParams const*, ...)" -> "mirp::fold_lds_kernel<0, true>"; other kernels lose their template arguments
    n = name.split("(")[0].replace("void ", "").strip()
    return n if n.startswith("mirp::fold_lds_kernel<") else re.sub(r"<.*>$", "", n)
FILL = "mirp::fold_lds_kernel<0, true>"      # the product's fill kernel of the default model (candidate-pool pass); <0, false> is the dense overflow pass
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
    for k in (FILL, "mirp::fold_lds_epilogue_kernel"):
        if k in acc:
            out.setdefault(re.sub(r"<.*>$", "", k.split("::")[1]) + "_sq_per_launch", {}).update({c: v / len(nl[k]) for c, v in acc[k].items()})
json.dump(out, open("profiles/%s_hbm_traffic_and_sq_pmc.json" % tag, "w"), indent=1, sort_keys=True)
shutil.copy(glob.glob("gpurun_out/%s_stats/*/*kernel_stats.csv" % tag)[0], "profiles/%s_kernel_stats.csv" % tag)
f185 = glob.glob("gpurun_out/%s_stats185/*/*kernel_stats.csv" % tag)
if f185:
    shutil.copy(f185[0], "profiles/%s_vienna185_kernel_stats.csv" % tag)
for sub, name in (("L400", "L400_kernel_stats.csv"), ("L400_185", "L400_vienna185_kernel_stats.csv")):
    fl = glob.glob("gpurun_out/%s_%s/*/*kernel_stats.csv" % (tag, sub))
    if fl:
        shutil.copy(fl[0], "profiles/%s_%s" % (tag, name))
for name in ("L400_counters.txt", "L400_vienna185_counters.txt"):
    fl = glob.glob("gpurun_out/%s_%s" % (tag, name))
    if fl:
        shutil.copy(fl[0], "profiles/%s_%s" % (tag, name))
# profiles/CURRENT.json: what bench.py reports as roofline.traffic / roofline.pipe_busy, with the commit the counters were collected at
import subprocess
stats = {r["Name"].split("(")[0].replace("void ", "").strip(): float(r["AverageNs"]) for r in csv.DictReader(open("profiles/%s_kernel_stats.csv" % tag))}
fill_ns = stats.get(FILL)
sq = out.get("fold_lds_kernel_sq_per_launch", {})
cur = {"source": "profiles/%s_hbm_traffic_and_sq_pmc.json + profiles/%s_kernel_stats.csv (rocprofv3 --pmc passes and --kernel-trace --stats of bench.py on MI355X; raw counter values)" % (tag, tag),
       "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None}
kf, ke = out["kernels"].get(FILL, {}), out["kernels"].get("mirp::fold_lds_epilogue_kernel", {})
cur["fold_fill_fetch_bytes_raw"] = 1024 * kf.get("FETCH_SIZE_KB_avg", 0.0)
cur["fold_fill_write_bytes"] = kf.get("write_bytes", 0.0)
cur["fold_fill_hbm_bytes_per_launch"] = cur["fold_fill_fetch_bytes_raw"] + cur["fold_fill_write_bytes"]
cur["fold_epilogue_fetch_bytes_raw"] = 1024 * ke.get("FETCH_SIZE_KB_avg", 0.0)
cur["fold_epilogue_write_bytes"] = ke.get("write_bytes", 0.0)
if fill_ns and sq:
    cyc = fill_ns * 2.4      # CU cycles per launch at 2.4 GHz
    n_cu = 256
    cur["fold_fill_avg_ns"] = fill_ns
    cur["fold_fill_pipe_busy"] = {
        "lds": sq.get("SQ_LDS_IDX_ACTIVE", 0.0) / (n_cu * cyc),                     # LDS-array cycles / (CUs x cycles)
        "lds_bank_conflict": sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / (n_cu * cyc),    # of which conflict replays
        "valu": sq.get("SQ_INSTS_VALU", 0.0) * 4.0 / (4 * n_cu * cyc),              # 4 cycles per wave64 instruction, 4 SIMDs per CU
        "salu": sq.get("SQ_INSTS_SALU", 0.0) / (n_cu * cyc),                        # one scalar issue per CU and cycle
        "insts_per_launch": {k: sq.get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")},
        "note": "busy fraction of each pipe over the fill kernel's launch: counter / (256 CUs x launch cycles at 2.4 GHz)"}
json.dump(cur, open("profiles/CURRENT.json", "w"), indent=1)
print(json.dumps(cur, indent=1))
print(json.dumps({k: v for k, v in out.items() if k.endswith("per_launch")}, indent=1))
print({k: v for k, v in out["kernels"].items() if "fold" in k})
