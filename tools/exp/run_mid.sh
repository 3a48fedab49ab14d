set -x
O=gpurun_out
python bench.py --config cfg3 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r06m_bench_cfg3.json
python bench.py --config cfg4 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r06m_bench_cfg4.json
python bench.py --scene clustered --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r06m_bench_cfg1_clustered.json
python bench.py --gpus 8 --config cfg4 --anchors 2000000 --exchange rs_ag --optimizer sharded --sparse-exchange --dry-run-ranks --steps 1 --warmup 1 > $O/r06m_dry_cfg4_sparse_sharded.log 2>&1
python bench.py --gpus 4 --config cfg3 --anchors 1000000 --dry-run-ranks --steps 1 --warmup 1 > $O/r06m_dry_cfg3.log 2>&1
python - <<'PY'
import json
for n in ("cfg3","cfg4","cfg1_clustered"):
    try:
        d=json.load(open(f"gpurun_out/r06m_bench_{n}.json")); print(n, d["value"], d["unit"], d["ms_per_step"], {k:v for k,v in list((d.get("kernel_ms_per_step") or d.get("kernel_ms") or {}).items())[:8]}, d["config"].get("largest_tile_entries"))
    except Exception as e: print(n, "failed", e)
PY
tail -3 $O/r06m_dry_cfg4_sparse_sharded.log; tail -3 $O/r06m_dry_cfg3.log
