#!/bin/bash
# Developer timing of the anchor gather kernels inside a configuration: tools/ab_gather.sh CONFIG lib [lib ...] ("base" = product)
CFG=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  D=$PWD/gpurun_out/abg_$$; R=$PWD
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $D -o s -- python3 $R/bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline > $D.log 2>&1)
  python3 tools/rocpd_summary.py $(find $D -name "*.db" | head -1) $D.txt x > /dev/null
  echo "$CFG $lib: $(grep anchor_gather $D.txt | awk '{print $1, $4}' | tr '\n' ' ') step $(grep '^{' $D.log | tail -1 | python3 -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],2))')"
  rm -rf $D $D.log $D.txt
done
