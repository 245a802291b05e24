cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in A B; do
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pm_$1_1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pm_$1_2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
break
done
