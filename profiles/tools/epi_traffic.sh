#!/bin/bash
# Dev tool: settles what the epilogue kernel reads from memory (task 6).  FETCH_SIZE and the raw L2 request counters of the epilogue for the product
# library and for the build without backtracks (make VARIANT=nobt VFLAGS=-DMIRP_X_EPI_NOBT: sweep + enumeration only, which reads the c and fML archives
# exactly once = a known byte count in this kernel's own access pattern, the calibration the guide asks for).   gpurun -- 'bash profiles/tools/epi_traffic.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for LIB in libmirprefer.so libmirprefer_vnobt.so; do
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_HIT TCC_MISS TCC_REQ"; do
    bash profiles/tools/pmc.sh mir-prefer_amd/$LIB epi_$(echo $LIB | tr -d '.')_$(echo $CTRS | cut -d' ' -f1) "$CTRS" epilogue
  done
done
