#!/bin/bash
# round 5, call G: the f16 UNet again: tests, time, LDS counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -s > $O/g_unet.log 2>&1; grep "f16 form\|passed\|failed\|PSNR\|1080p" $O/g_unet.log | head
rm -rf /tmp/prof_unet
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet -o unet -- python $GRAFT_REPO_ROOT/tools/unet_bench.py 6 f16 > $O/g_unet_bench.log 2>&1); tail -1 $O/g_unet_bench.log
python - <<'PY'
import csv,glob
rows=[]
for f in glob.glob("/tmp/prof_unet/**/*kernel_trace.csv", recursive=True):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
convs=[r for r in rows if "conv3x3" in r[2] or "image_inputs" in r[2]]
last=convs[-17:]
for s,e,k in last:
    print(f"{(e-s)/1e3:8.1f} us  {k.split('(')[0][-40:]}")
print("sum", sum(e-s for s,e,_ in last)/1e3, "us; wall", (last[-1][1]-last[0][0])/1e3)
PY
i=0
for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/pmc_unet$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/pmc_unet$i -o unet -- python $GRAFT_REPO_ROOT/tools/unet_bench.py 2 f16 > $O/g_pmc$i.log 2>&1)
done
python3 tools/summarize_pmc.py /tmp pmc_unet x 2>/dev/null | grep "k_conv3x3_h" | cut -c1-170 > $O/g_unet_f16_pmc.txt
grep "<4, 4>\|<1, 4>" $O/g_unet_f16_pmc.txt
