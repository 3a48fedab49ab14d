#!/bin/bash
# Developer A/B: run bench.py once per variant library given on the command line.
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['ms_per_step'],3), d['kernel_ms'])"
done
