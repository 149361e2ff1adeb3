#!/bin/bash
# round 5, call F: where the f16 UNet's time goes (counters)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/pmc_unet$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/pmc_unet$i -o unet -- python $GRAFT_REPO_ROOT/tools/unet_bench.py 2 f16 > $O/f_pmc$i.log 2>&1)
done
python3 tools/summarize_pmc.py /tmp pmc_unet x 2>/dev/null | grep "k_conv3x3_h" | cut -c1-170 > $O/f_unet_f16_pmc.txt
cat $O/f_unet_f16_pmc.txt | grep "<4, 4>\|<1, 4>\|<2, 4>"
