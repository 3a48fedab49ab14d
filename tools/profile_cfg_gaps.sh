#!/bin/bash
# Run on the GPU box (through gpurun): idle gaps of one steady-state step of a bench.py configuration (kernel trace as csv).
# usage: tools/profile_cfg_gaps.sh TAG CONFIG [min gap us] [extra bench.py arguments]  -> gpurun_out/TAG_step_gaps.txt
TAG=$1; CFG=$2; MIN=${3:-5}; shift 3 2>/dev/null || shift $#
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_$TAG -o trace -- python3 $REPO/bench.py --config $CFG --steps 10 --warmup 5 --no-cpu-baseline "$@" > $OUT/gaps_$TAG.log 2>&1
cd $REPO
CSV=$(find $OUT/gaps_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/step_gaps_cfg.py $CSV $MIN > $OUT/${TAG}_step_gaps.txt 2>&1
tail -60 $OUT/${TAG}_step_gaps.txt
rm -rf $OUT/gaps_$TAG
