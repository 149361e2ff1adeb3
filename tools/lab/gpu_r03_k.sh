#!/bin/bash
# round 3: (1) the vector ALU's issue cost per instruction kind, (2) SQ counters of K2 for the 4-wide and the 8-wide walk
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03k
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 tools/_build/valu_bench > $OUT/valu_bench.txt 2>&1; echo "valu_bench exit $?"; cat $OUT/valu_bench.txt
export PMC_STEPS=20
for W in 4 8; do
  RAYHIP_BVH_WIDTH=$W bash tools/gpu_pmc.sh r03k/w$W bistro "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" > $OUT/pmc_w$W.txt 2>&1
  grep "refill\|k_trace_shadow\|k_scatter\|k_surface\|k_light_pick" $OUT/pmc_w$W.txt | head -80
done
