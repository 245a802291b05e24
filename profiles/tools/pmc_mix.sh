#!/bin/bash
# Dev tool: instruction mix of the fill kernel by class (rocprofv3 --pmc, two passes):  bash profiles/tools/pmc_mix.sh   (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SENDMSG" "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED SQ_WAVE_CYCLES"; do
  rm -rf gpurun_out/mix; timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d gpurun_out/mix -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --no-ingest --no-configs > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/mix/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "fold_lds_kernel<0, true>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(agg): print("mix %-28s %.4g per launch" % (k, agg[k] / max(1, len(n[k]))))
PY
done
