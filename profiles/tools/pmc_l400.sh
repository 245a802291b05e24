#!/bin/bash
# Dev tool: rocprofv3 PMC passes over the generic fold path at PRECURSOR_LEN = 400 (profiles/tools/l400_time.py); prints the generic fill kernel's counters per launch.
#   gpurun -- 'bash profiles/tools/pmc_l400.sh [library] [model]'
LIB=${1:-mir-prefer_amd/libmirprefer.so}; MODEL=${2:-vienna-2.1.2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MIRP_LIB=$PWD/$LIB
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" "TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  TAG=l400_$(echo $CTRS | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$TAG
  timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_$TAG -- python3 profiles/tools/l400_time.py 400 $MODEL > gpurun_out/pmc_$TAG.log 2>&1
  python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if ("fold_generic_kernel<1>" in name or "fold185_kernel<1>" in name or "ILi1EE" in name):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(tag, {k: "%.4g (x%d launches)" % (v, len(n[k])) for k, v in sorted(agg.items())})
PY
done
