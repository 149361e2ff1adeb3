#!/bin/bash
# round 6, call E: the whole GPU suite on the fused shade form + counters of its kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v) for k,v in d['stage_us_per_spp'].items()})"
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest exit $?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/gputest.log | cut -c1-220 | tail -30
cd /tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc$i -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $O/pmc$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_pmc.py $O pmc x | grep "shade::" | cut -c1-200 > $O/pmc_shade_summary.txt
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
cat $O/pmc_shade_summary.txt | grep -v "emissive\|<true, true" 
