#!/bin/bash
# Run on the GPU box (through gpurun): ONE training step of a configuration under rocprofv3 --marker-trace --kernel-trace with the
# stage markers on (SPLATCO_MARKERS=1), summarised per stage into gpurun_out/TAG_marker_trace.txt (tools/marker_summary.py).
# usage: tools/profile_markers.sh TAG [CONFIG]       (no --pmc: a trace-only run)
TAG=${1:-r}; CFG=${2:-cfg3}
REPO=$PWD; OUT=$REPO/gpurun_out
export TMPDIR=/tmp SPLATCO_MARKERS=1
mkdir -p $OUT; rm -rf $OUT/mk_$TAG
(cd /tmp && timeout -k 5 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $OUT/mk_$TAG -o mk -- python3 $REPO/bench.py --config $CFG --steps 2 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_marker_bench.log 2>&1)
python3 tools/marker_summary.py $OUT/mk_$TAG $OUT/${TAG}_marker_trace.txt "bench.py --config $CFG --steps 2 --warmup 2 (SPLATCO_MARKERS=1, rocprofv3 --marker-trace --kernel-trace)"
rm -rf $OUT/mk_$TAG
