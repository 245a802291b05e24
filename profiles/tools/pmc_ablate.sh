#!/bin/bash
# Dev tool: SQ instruction counters of the fill kernel with phases ablated (MIRP_FOLD_DEBUG), to see where the instructions are issued.
#   gpurun -- 'bash profiles/tools/pmc_ablate.sh "0 1 2 3"'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for F in $1; do
  export MIRP_FOLD_DEBUG=$F
  rm -rf gpurun_out/pa_$F
  timeout 120 rocprofv3 --kernel-trace --pmc ${PMC_CTRS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY} --output-format csv -d gpurun_out/pa_$F -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python3 - "$F" <<'PY'
import csv, glob, collections, sys
f = glob.glob("gpurun_out/pa_%s/*/*counter_collection.csv" % sys.argv[1])[0]
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if "fold_lds_kernel" in r["Kernel_Name"] and "epilogue" not in r["Kernel_Name"] and ("true>" in r["Kernel_Name"] or "<1" in r["Kernel_Name"]):
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("flags", sys.argv[1], {k: "%.3g" % (v / len(n[k])) for k, v in sorted(agg.items())})
PY
done
