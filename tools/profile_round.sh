#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + three PMC passes of bench.py cfg1.
# usage: tools/profile_round.sh TAG [pmc-only]  -> gpurun_out/prof_TAG (rocpd db), gpurun_out/pmc_TAG_{fetch,write,sq1,sq2}
TAG=${1:-r}
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp
B="$REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 $BENCH_ARGS"      # BENCH_ARGS: e.g. "--sigma-scale 0.1"
# every pass under its own timeout: a failed counter request leaves rocprofv3 hanging in its signal handler;
# FETCH_SIZE and WRITE_SIZE do not fit in one pass ("exceeds the capabilities of the hardware to collect")
T="timeout -k 5 240"
if [ "$2" != "pmc-only" ]; then
$T rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o stats -- python3 $B > $OUT/bench_$TAG.log 2>&1
fi
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch -o fetch -- python3 $B > $OUT/pmc_$TAG.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write -o write -- python3 $B >> $OUT/pmc_$TAG.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_${TAG}_sq1 -o sq1 -- python3 $B >> $OUT/pmc_$TAG.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_sq2 -o sq2 -- python3 $B >> $OUT/pmc_$TAG.log 2>&1
cd $REPO
tail -1 $OUT/bench_$TAG.log
find $OUT/prof_$TAG -name "*.db" | head -2
