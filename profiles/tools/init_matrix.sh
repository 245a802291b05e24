#!/bin/bash
# Dev tool (GPU box, repo root): HIP start-up time (profiles/tools/bin/ctx_probe bare) under environment settings that might shorten it, three runs each,
# and one run under the LD_PRELOAD call timer (sysprobe.so).
P=profiles/tools/bin/ctx_probe
run() { echo "## $*"; for i in 1 2 3; do env "$@" $P bare 2>&1 | grep -E "hipInit|hipStreamCreate|total before" | awk '{printf "%s ", $(NF-1)} END {print ""}'; done; }
run A=1
run HSA_ENABLE_SDMA=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=1
run ROCR_VISIBLE_DEVICES=0
run HSA_NO_SCRATCH_RECLAIM=1
run HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0
run HSA_ENABLE_SDMA=0 HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=1
run ROCPROFILER_REGISTER_DISABLE=1
run HSA_OVERRIDE_CPU_AFFINITY_DEBUG=0
run HSA_DISABLE_FRAGMENT_ALLOCATOR=1
run HSA_ENABLE_DEBUG=0 HSA_TOOLS_LIB=
echo "## sysprobe"
LD_PRELOAD=profiles/tools/bin/sysprobe.so $P bare 2>&1 | grep -E "sysprobe|hipInit|hipStream"
nproc; ls /sys/class/kfd/kfd/topology/nodes | wc -l; ls /sys/devices/system/cpu | grep -c "^cpu[0-9]"
