#!/bin/bash
# round 4: the pooled kernel against the refill kernel -- what the stack depth costs, and the SQ counters of both
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04b
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
export RT_VARIANTS="$(cat tools/variants_pool3_r04.json)"
RT_BATCH=64 timeout 900 python tools/variants.py run bistro 64 > $OUT/variants_pool3_bistro64.txt 2>&1; echo "variants exit $?"
grep -v "^    " $OUT/variants_pool3_bistro64.txt | cut -c1-200
export PMC_STEPS=1
for m in 3 4; do
  RAYHIP_REFILL=$m timeout 600 bash tools/gpu_pmc.sh r04b/mode$m bistro "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" > $OUT/pmc_mode$m.txt 2>&1
  grep "k_trace_closest_pool\|refillILi4ELi40\|refill<4, 40>" $OUT/pmc_mode$m.txt | awk '{print $(NF-6), $(NF-4), $(NF-2), $NF}' | head -14
done
du -sh $OUT
