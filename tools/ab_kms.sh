#!/bin/bash
# Developer A/B: bench.py cfg1 per-kernel milliseconds for each variant library ("base" = product library)
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$lib', round(d['value'],1), round(d['ms_per_step'],4), 'fwd', k.get('blend_forward_kernel'), 'bwd', k.get('blend_backward_kernel'))"
done
