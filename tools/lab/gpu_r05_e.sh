#!/bin/bash
# round 5, call E: the f16 UNet: tests, then the per-pass kernel times of both forms at 1080p
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -s -k f16 > $O/e_unet.log 2>&1; grep -v "^Extends\|^Spatial" $O/e_unet.log | grep "pass \|f16\|passed\|failed\|PSNR" | head -60
for form in f16; do
  rm -rf /tmp/prof_unet_$form
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet_$form -o unet -- python $GRAFT_REPO_ROOT/tools/unet_bench.py 6 $form > $O/e_unet_bench_$form.log 2>&1)
  tail -1 $O/e_unet_bench_$form.log
  cp $(find /tmp/prof_unet_$form -name "*kernel_stats.csv" | head -1) $O/e_unet_kernel_stats_$form.csv
  python - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("/tmp/prof_unet_$form/**/*kernel_trace.csv", recursive=True):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
convs=[r for r in rows if "conv3x3" in r[2] or "image_inputs" in r[2]]
# the last frame: 16 convs + 2 image_inputs
last=convs[-18:]
for s,e,k in last:
    print(f"{(e-s)/1e3:8.1f} us  {k.split('(')[0][-40:]}")
print("sum", sum(e-s for s,e,_ in last)/1e3, "us; wall", (last[-1][1]-last[0][0])/1e3)
PY
done
