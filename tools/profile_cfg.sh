#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of one bench.py configuration.
# usage: tools/profile_cfg.sh TAG CONFIG [extra bench args]  -> gpurun_out/prof_TAG (rocpd db), gpurun_out/bench_TAG.log,
#        profiles-ready summary gpurun_out/TAG_kernel_stats.txt
TAG=$1; CFG=$2; shift 2
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o stats -- python3 $REPO/bench.py --config $CFG --no-cpu-baseline "$@" > $OUT/bench_$TAG.log 2>&1
cd $REPO
tail -1 $OUT/bench_$TAG.log
DB=$(find $OUT/prof_$TAG -name "*.db" | head -1)
python3 tools/rocpd_summary.py $DB $OUT/${TAG}_kernel_stats.txt "bench.py --config $CFG $* (rocprofv3 --kernel-trace --stats)" | head -60
