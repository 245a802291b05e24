#!/bin/bash
# Dev tool: texture-addresser / L1 counters (TA_BUSY, TCP accesses and stalls, TD_BUSY) of the generic fill kernel at PRECURSOR_LEN = 400, one rocprofv3 --pmc pass per group.
#   gpurun -- 'bash profiles/tools/pmc_l400_ta.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MIRP_LIB=$PWD/mir-prefer_amd/libmirprefer.so
for CTRS in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  TAG=l400b_$(echo $CTRS | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$TAG
  timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_$TAG -- python3 profiles/tools/l400_time.py 400 > gpurun_out/pmc_$TAG.log 2>&1
  tail -2 gpurun_out/pmc_$TAG.log | head -1
  python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for f in glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if ("fold_generic_kernel<1>" in name or "ILi1EE" in name):
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(tag, {k: "%.4g (x%d)" % (v, len(n[k])) for k, v in sorted(agg.items())})
PY
done
