#!/bin/bash
# Developer A/B on the anchor path's kernels: tools/ab_heads.sh CONFIG lib [lib ...]   ("base" = the product library)
CFG=$1; shift
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$PWD/$lib; fi
  python bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$CFG $lib', round(d['ms_per_step'],2), {n: round(k.get(n,0),3) for n in ('mlp_heads_kernel','mlp_heads_backward_kernel','norm_linear_kernels','norm_linear_backward_kernels','expand_kernel','expand_backward_kernel','triplane_forward_kernel','plane_sample_backward_kernels')})"
done
