#!/bin/bash
# Developer A/B: FETCH_SIZE / WRITE_SIZE per kernel (KB per launch) of bench.py cfg1 for each library given ("base" = product)
export TMPDIR=/tmp
REPO=$PWD
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$REPO/$lib; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abpmc; (cd /tmp && timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/abpmc -o x -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg2 > /dev/null 2>&1)
    python3 - "$lib" $c <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("/tmp/abpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("scr::", "").split("<")[0]
        if r["Counter_Name"] == sys.argv[2]:
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print(sys.argv[1], sys.argv[2], {k: round(v[0] / v[1] / 1024, 1) for k, v in acc.items() if k in ("scatter_kernel", "tile_sort_wave_kernel", "preprocess_kernel", "plan_scan_kernel")}, "MB per launch")
PY
  done
done
