#!/bin/bash
# round 4: the UNet passes with software-pipelined staging -- parity, time, per-pass kernel trace, matrix-core busy cycles
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04f
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp RAY_AMD_CACHE=/tmp/ray_amd_cache
cd $REPO
timeout 300 python -m pytest tests/test_gpu_unet.py -m gpu -q -x -s > $OUT/gputest_unet.log 2>&1; echo "pytest exit $?"; grep "UNet 1080p\|passed\|failed" $OUT/gputest_unet.log | tail -4
timeout 120 python tools/unet_bench.py 10 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o unet -- python $REPO/tools/unet_bench.py 5 > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/unet_kernel_stats.csv 2>/dev/null; head -12 $OUT/unet_kernel_stats.csv | cut -c1-160
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o unet -- python $REPO/tools/unet_bench.py 3 > $OUT/pmc.log 2>&1
python3 $REPO/tools/summarize_pmc.py $OUT pmc x 2>/dev/null | grep "k_conv3x3" | cut -c1-200 > $OUT/unet_pmc_summary.txt; cat $OUT/unet_pmc_summary.txt | head -30
find $OUT -name '*.csv' -size +6M -delete; find $OUT -name '*.db' -delete
