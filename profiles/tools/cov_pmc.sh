#!/bin/bash
# Dev tool: HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and durations of the coverage-stage kernels at config[1] and at a config[4] rank
# shard (profiles/tools/cov_time.py), fused scan and atomic path.   gpurun -- 'bash profiles/tools/cov_pmc.sh r3_b'
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in 1 0; do
  export MIRP_COV_FUSED=$mode
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_cov${mode}_stats -- python3 profiles/tools/cov_time.py > gpurun_out/${TAG}_cov${mode}.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_cov${mode}_fetch -- python3 profiles/tools/cov_time.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_cov${mode}_write -- python3 profiles/tools/cov_time.py > /dev/null 2>&1
  grep coverage gpurun_out/${TAG}_cov${mode}.log
done
