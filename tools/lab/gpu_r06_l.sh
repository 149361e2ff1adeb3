#!/bin/bash
# round 6, call L: the other BASELINE workloads on the new shade form against the three-kernel form; the UNet's roofline table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06l; rm -rf $O; mkdir -p $O
for w in cornell principled sponza bistro_tex; do
 for split in 29 13; do
  RAYHIP_SHADE_SPLIT=$split timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_${w}_$split.json 2> $O/b.err
  python3 -c "
import json; d=json.load(open('$O/bench_${w}_$split.json')); print('$w split $split', round(d['value'],1), 'Msamples/s', {k: round(v) for k,v in d['stage_us_per_spp'].items()})" | tee -a $O/forms.txt
 done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/unet_trace -o u -- python $GRAFT_REPO_ROOT/tools/unet_bench.py 6 f16 > $O/unet_bench.log 2>&1; tail -1 $O/unet_bench.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/unet_$c -o u -- python $GRAFT_REPO_ROOT/tools/unet_bench.py 2 f16 > $O/unet_$c.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/unet_roofline.py $O/unet_trace $O/unet_FETCH_SIZE $O/unet_WRITE_SIZE > $O/unet_f16_roofline.txt 2>&1; cat $O/unet_f16_roofline.txt
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
