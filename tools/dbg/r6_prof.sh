#!/bin/bash
# round 6 profile session: ubenches, the round's profile set (tools/gpu_prof.sh), the list of counters this GPU has
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
make -C oracle > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/gj_rows.hip -o /tmp/gj_rows && /tmp/gj_rows | tee gpurun_out/r06_ubench_gj_rows.txt
(rocprofv3-avail list 2>/dev/null || rocprofv3 -L 2>/dev/null) | grep -oE '\b(SQ|TCP|TCC|TA|TD)_[A-Z0-9_]+' | sort -u > gpurun_out/r06_counters_available.txt; wc -l gpurun_out/r06_counters_available.txt
export PMC_EXTRA_SETS="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS;TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_HIT_sum TCC_MISS_sum"
ROUND=${ROUND:-r06_a} bash tools/gpu_prof.sh
