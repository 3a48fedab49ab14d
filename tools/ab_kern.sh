#!/bin/bash
# Developer A/B of named kernels inside a configuration under rocprofv3 --kernel-trace --stats:
# tools/ab_kern.sh CONFIG "grep-pattern" lib [lib ...]   ("base" = product library)
CFG=$1; PAT=$2; shift 2
export TMPDIR=/tmp
R=$PWD
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$R/$lib; fi
  D=$R/gpurun_out/abk_$$
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $D -o s -- python3 $R/bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline > $D.log 2>&1)
  python3 tools/rocpd_summary.py $(find $D -name "*.db" | head -1) $D.txt x > /dev/null
  echo "$CFG $(basename $lib): $(grep -E "$PAT" $D.txt | awk '{print $1, $4}' | sed 's/scr:://' | tr '\n' ' ')"
  rm -rf $D $D.log $D.txt
done
