#!/bin/bash
# Run on the GPU box: the round's final bench lines with the counter tables installed (default line incl. cfg2 + CPU baseline; cfg3; cfg4).
O=gpurun_out; T=${1:-r06z}
python bench.py > $O/${T}_bench_cfg1.json 2> $O/${T}_bench_cfg1.err
python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${T}_bench_cfg3.json
python bench.py --config cfg4 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${T}_bench_cfg4.json
python - <<PY
import json
d=json.load(open('$O/${T}_bench_cfg1.json'))
print({k:d[k] for k in ('value','ms_per_step','effective_warmup_steps','ramp_ms_per_step','kernel_sum_ms')})
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','traffic','avg_launch_ms')}, r['valu'].get('valu_frac'), r['valu'].get('vs_mix_probe'))
print(d['cpu_baseline']['value'], d['cfg2']['ms_per_step'], d['cfg2']['roofline']['frac'], d['cfg2']['roofline'].get('traffic'))
for n in ('cfg3','cfg4'):
    e=json.load(open('$O/${T}_bench_'+n+'.json')); print(n, e['value'], e['unit'], e['ms_per_step'])
PY
