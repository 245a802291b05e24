cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 0 4 8 2 16 31; do
MIRP_FOLD_DEBUG=$f timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/abl_$f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
