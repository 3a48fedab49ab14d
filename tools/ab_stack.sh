#!/bin/bash
# Developer A/B: cfg2 with the level-0 grids stacked (default) and separate
for v in 1 0 1 0; do
  SPLATCO_STACK_LEVEL0=$v python bench.py --config ${1:-cfg2} --steps 5 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('stack=$v', round(d['ms_per_step'],2), {n: round(k.get(n,0),3) for n in ('triplane_forward_kernel','plane_sample_backward_kernels','plane_attention_kernels','norm_linear_kernels','norm_linear_backward_kernels')})"
done
