#!/bin/bash
# Developer A/B: per-kernel-class ms per step of a configuration, product library against a variant, and the differences.
# usage: tools/ab_diff.sh CONFIG variant.so [extra bench.py arguments]
CFG=$1; VAR=$2; shift 2
run() { python bench.py --config $CFG --steps 6 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep '^{"metric"' | tail -1; }
unset SPLATCO_RASTER_LIB; run "$@" > /tmp/ab_base.json
SPLATCO_RASTER_LIB=$PWD/$VAR run "$@" > /tmp/ab_var.json
python - <<'PY'
import json
a = json.loads(open('/tmp/ab_base.json').read()); b = json.loads(open('/tmp/ab_var.json').read())
ka, kb = a.get('kernel_ms_per_step') or a.get('kernel_ms'), b.get('kernel_ms_per_step') or b.get('kernel_ms')
print(f"step: product {a['ms_per_step']:.3f} ms, variant {b['ms_per_step']:.3f} ms; tracked kernels {sum(ka.values()):.3f} / {sum(kb.values()):.3f}; "
      f"device allocations in the timed region {a.get('device_allocs_in_timed_region')} / {b.get('device_allocs_in_timed_region')}; reserved {a.get('reserved_gib')} / {b.get('reserved_gib')} GiB")
for k in sorted(set(ka) | set(kb), key=lambda k: -abs(ka.get(k, 0) - kb.get(k, 0)))[:10]:
    print(f"  {ka.get(k, 0) - kb.get(k, 0):+.3f}  {k:40s} product {ka.get(k, 0):.3f}  variant {kb.get(k, 0):.3f}")
PY
