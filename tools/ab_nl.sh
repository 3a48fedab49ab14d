#!/bin/bash
# Developer A/B on the BatchNorm-Linear kernels: tools/ab_nl.sh CONFIG lib [lib ...]
CFG=$1; shift
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  python bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$CFG', '$lib'.split('/')[-1], round(d['ms_per_step'],2), {n: round(k.get(n,0),3) for n in ('norm_linear_kernels','norm_linear_backward_kernels')})"
done
