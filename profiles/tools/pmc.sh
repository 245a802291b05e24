#!/bin/bash
# Dev tool: rocprofv3 PMC pass over the bench workload's fold (profiles/tools/ab_time.py) for one library; prints the fill kernel's counters per launch.
#   gpurun -- 'bash profiles/tools/pmc.sh mir-prefer_amd/libmirprefer.so tag "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ..."'
LIB=$1; TAG=$2; CTRS=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MIRP_LIB=$LIB
rm -rf gpurun_out/pmc_$TAG
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_$TAG -- python3 profiles/tools/ab_time.py > gpurun_out/pmc_$TAG.log 2>&1
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "fold_lds_kernel" in r["Kernel_Name"] and "epilogue" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(tag, {k: "%.4g" % (v / len(n[k])) for k, v in sorted(agg.items())})
PY
