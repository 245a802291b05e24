#!/bin/bash
# Dev tool: rocprofv3 PMC pass over the bench workload's fold (profiles/tools/ab_time.py) for one library; prints the fill kernel's counters per launch.
#   gpurun -- 'bash profiles/tools/pmc.sh mir-prefer_amd/libmirprefer.so tag "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ..." [kernel-name substring, default: the fill kernel]'
LIB=$1; TAG=$2; CTRS=$3; KERN=${4:-fill}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MIRP_LIB=$LIB
rm -rf gpurun_out/pmc_$TAG
timeout 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_$TAG -- python3 profiles/tools/ab_time.py > gpurun_out/pmc_$TAG.log 2>&1
python3 - "$TAG" "$KERN" <<'PY'
import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if ("fold_lds_kernel" in name and "epilogue" not in name and ("true>" in name or "<1" in name)) if kern == "fill" else (kern in name):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(tag, {k: "%.4g" % (v / len(n[k])) for k, v in sorted(agg.items())})
PY
