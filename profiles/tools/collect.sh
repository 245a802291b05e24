#!/bin/bash
# Profiling recipe used for the committed summaries under profiles/ (run on the GPU box from the repo root):
#   gpurun -- 'bash profiles/tools/collect.sh r1_e'
# 1. rocprofv3 kernel trace + stats of the default bench command; 2./3. HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes);
# 4./5. SQ instruction / wait / LDS counters of the dominant kernel.
TAG=${1:-rX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-ingest --no-configs > gpurun_out/${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --no-ingest --no-configs > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --no-ingest --no-configs > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${TAG}_sq -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --no-ingest --no-configs > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d gpurun_out/${TAG}_sq2 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-e2e --no-ingest --no-configs > /dev/null 2>&1
# 6. the vienna-1.8.5 model (fold_lds_kernel<1> + fold185_lds_epilogue_kernel): kernel stats of the same workload
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats185 -- python3 bench.py --steps 5 --warmup 1 --fold-model vienna-1.8.5 --no-cpu-baseline --no-e2e --no-ingest --no-configs > gpurun_out/${TAG}_stats185.log 2>&1
tail -1 gpurun_out/${TAG}_stats.log | cut -c1-200
tail -1 gpurun_out/${TAG}_stats185.log | cut -c1-200
# 7. the generic path (PRECURSOR_LEN = 400: fold_generic_kernel / fold185_kernel over config[1]'s 19,686 windows, two folds each): kernel stats and the fill kernel's counters
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_L400 -- python3 profiles/tools/l400_time.py 400 > gpurun_out/${TAG}_L400.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_L400_185 -- python3 profiles/tools/l400_time.py 400 vienna-1.8.5 > gpurun_out/${TAG}_L400_185.log 2>&1
( tail -2 gpurun_out/${TAG}_L400.log; echo "counters of fold_generic_kernel<1>, sums over the launches of two folds (rocprofv3 --pmc, one pass per group):"; bash profiles/tools/pmc_l400.sh 2>&1 | grep "^l400" ) > gpurun_out/${TAG}_L400_counters.txt
( tail -2 gpurun_out/${TAG}_L400_185.log; echo "counters of fold185_kernel<1>, sums over the launches of two folds:"; bash profiles/tools/pmc_l400.sh mir-prefer_amd/libmirprefer.so vienna-1.8.5 2>&1 | grep "^l400" ) > gpurun_out/${TAG}_L400_vienna185_counters.txt
tail -3 gpurun_out/${TAG}_L400_counters.txt | cut -c1-200
