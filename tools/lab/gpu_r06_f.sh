#!/bin/bash
# round 6, call F: the Bistro-class street on the reference's asset mesh (baked / instanced, plain SAH / spatial splits): bench lines with the visit
# census, parity at 1080p / 64 spp against RendererRef, counter profiles (FETCH / WRITE / VALU) of the two main variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
O=$GRAFT_REPO_ROOT/gpurun_out/r06f; rm -rf $O; mkdir -p $O
for w in bistro_assets bistro_assets_inst bistro_assets_sbvh bistro_assets_inst_sbvh bistro; do
  extra="--no-cpu-baseline"; [ $w = bistro_assets ] && extra=""
  timeout 900 python bench.py --workload $w --steps 6 --warmup 2 $extra > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?"
  python3 -c "
import json; d=json.load(open('$O/bench_$w.json')); a=d['roofline']['algorithmic']; print('$w', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms', {k: round(v) for k,v in d['stage_us_per_spp'].items()}, 'tris', d['config']['unique_tris'], 'build', round(d['scene_build_s'],1), '| per ray: tlas', round(a['tlas_nodes_per_ray'],2), 'wide', round(a['wide_nodes_per_ray'],2), 'tris', round(a['tris_per_ray'],2), 'inst', round(a['instances_per_ray'],2), 'rays/sample', round(d['roofline']['rays_per_sample'],2), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('pass'))"
done
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -s -k "asset_street" > $O/parity_assets.log 2>&1; echo "pytest exit $?"
grep "spp\|passed\|failed" $O/parity_assets.log | cut -c1-250
cd /tmp
cp $GRAFT_REPO_ROOT/profiles/r05/k2_traffic.json $O/k2_traffic.json
for w in bistro_assets bistro_assets_inst; do
  steps=4
  for tag in FETCH_SIZE WRITE_SIZE VALU; do
    pmc=$tag; [ $tag = VALU ] && pmc="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
    timeout 600 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pmc_${w}_${tag} -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps $steps --warmup 1 --no-cpu-baseline > $O/pmc_${w}_${tag}.log 2>&1
  done
  python3 $GRAFT_REPO_ROOT/tools/k2_traffic.py $O/k2_traffic.json $w $steps 1 64 64 $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/kernel_hbm_${w}.txt $O/pmc_${w}_VALU > $O/k2_traffic_$w.log 2>&1
  head -1 $O/k2_traffic_$w.log | cut -c1-400
  head -16 $O/kernel_hbm_${w}.txt | cut -c1-170
  find $O -name '*.csv' -size +4M -delete; find $O -name '*.db' -delete
done
