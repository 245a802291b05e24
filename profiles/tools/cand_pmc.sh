#!/bin/bash
# Dev tool: rocprofv3 PMC pass over the candidate stage at a config[4] rank shard (profiles/tools/cand_shard.py); prints a kernel's counters per launch.
#   gpurun -- 'bash profiles/tools/cand_pmc.sh tag "SQ_WAVE_CYCLES SQ_BUSY_CYCLES ..." [kernel-name substring, default cov_scan_kernel]'
TAG=$1; CTRS=$2; KERN=${3:-cov_scan_kernel}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/cpmc_$TAG
timeout 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/cpmc_$TAG -- python3 profiles/tools/cand_shard.py > gpurun_out/cpmc_$TAG.log 2>&1
python3 - "$TAG" "$KERN" <<'PY'
import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
for f in glob.glob("gpurun_out/cpmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(tag, {k: "%.4g" % (v / len(n[k])) for k, v in sorted(agg.items())})
PY
