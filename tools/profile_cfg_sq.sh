#!/bin/bash
# Run on the GPU box (through gpurun): SQ issue counters of one anchor configuration, per kernel.
# usage: tools/profile_cfg_sq.sh TAG CONFIG -> gpurun_out/TAG_CONFIG_sq_table.txt
TAG=$1; CFG=$2
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
B="$REPO/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline"
i=0
for c in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/sq_${TAG}_${CFG}_$i -o p$i -- python3 $B > $OUT/sq_${TAG}_${CFG}_$i.log 2>&1
done
cd $REPO
python3 tools/pmc_table.py $OUT/${TAG}_${CFG}_sq_table.txt $OUT/sq_${TAG}_${CFG}_[1-4] | cut -c1-400 | head -60
rm -rf $OUT/sq_${TAG}_${CFG}_[1-4]
