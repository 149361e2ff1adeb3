#!/bin/bash
# round 3: SQ / TCC counters of the final kernels (20-layer passes), per kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
export PMC_STEPS=20
timeout 280 bash tools/gpu_pmc.sh r03pmc bistro "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" > gpurun_out/r03pmc.txt 2>&1
grep -c . gpurun_out/r03pmc.txt; grep "refill<4, 40>" gpurun_out/r03pmc.txt | awk '{print $4,$5,$6,$7,$8}' | head -20
