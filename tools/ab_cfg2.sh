#!/bin/bash
# Developer A/B: bench.py --config cfg2 once per library ("base" = product): ms per step and the per-class kernel times.
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  python bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step') or d.get('kernel_ms') or {}
print('$lib', round(d['ms_per_step'],3), {a: b for a, b in list(k.items())[:14]})"
done
