#!/bin/bash
# Per-kernel times of the candidate stage at the size of a config[4] rank shard: gpurun -- 'bash profiles/tools/cand_shard_trace.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/cs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs -- python3 profiles/tools/cand_shard.py > gpurun_out/cs.log 2>&1
grep coverage_ms gpurun_out/cs.log | tail -2
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/cs/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-64s calls %5s avg %9.1f us  per stage %8.1f us" % (r["Name"].replace("void ", "").replace("mirp::", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e3))
PY
