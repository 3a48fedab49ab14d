#!/bin/bash
# Developer A/B at 4K (32400 tiles): LDS histogram (160 KB) vs the global-atomic fallback
for lib in base splatco_amd/csrc/exp/lib_hist16k.so; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  echo $lib; python tools/kernel_breakdown.py 1000000 3840 2160
done
