cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_CACHE|INSTR)[A-Z_0-9]*)\b" | sort -u | head -40
timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic1 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/ic1.log 2>&1
tail -2 gpurun_out/ic1.log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/ic1/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "fold_lds_kernel" in r["Kernel_Name"] and "epilogue" not in r["Kernel_Name"] and ("true>" in r["Kernel_Name"] or "<1" in r["Kernel_Name"]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in agg: print(k, agg[k] / max(1, n[k]), n[k])
PY
