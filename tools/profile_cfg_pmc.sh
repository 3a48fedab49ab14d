#!/bin/bash
# Run on the GPU box (through gpurun): HBM traffic counters of one anchor configuration, reduced on the box.
# usage: tools/profile_cfg_pmc.sh TAG CONFIG -> gpurun_out/TAG_CONFIG_pmc_summary.txt, gpurun_out/hbm_traffic_CONFIG_TAG.json
TAG=$1; CFG=$2
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
B="$REPO/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}_${CFG}_$c -o $c -- python3 $B > $OUT/pmc_${TAG}_${CFG}_$c.log 2>&1
done
cd $REPO
python3 tools/pmc_cfg_summary.py $OUT/${TAG}_${CFG}_pmc_summary.txt $OUT/hbm_traffic_${CFG}_$TAG.json $OUT/pmc_${TAG}_${CFG}_FETCH_SIZE $OUT/pmc_${TAG}_${CFG}_WRITE_SIZE | head -50
rm -rf $OUT/pmc_${TAG}_${CFG}_FETCH_SIZE $OUT/pmc_${TAG}_${CFG}_WRITE_SIZE
