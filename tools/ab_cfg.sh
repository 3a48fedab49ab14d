#!/bin/bash
# Developer A/B on an anchor configuration: tools/ab_cfg.sh CONFIG lib [lib ...]   ("base" = the product library)
CFG=$1; shift
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  python bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$CFG $lib', round(d['ms_per_step'],2), {n: round(k.get(n,0),3) for n in ('scatter_kernel','tile_sort_kernel','blend_forward_kernel','blend_backward_kernel','preprocess_backward_kernel','preprocess_kernel')})"
done
