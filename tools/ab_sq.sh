#!/bin/bash
# Developer A/B: SQ counters of ONE kernel of bench.py cfg1 for each library given ("base" = product), averages per launch.
# usage: tools/ab_sq.sh KERNEL lib [lib ...]    (three PMC passes per library, --kernel-trace only)
export TMPDIR=/tmp
REPO=$PWD
KERN=$1; shift
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$REPO/$lib; fi
  rm -rf /tmp/absq
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
              "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
              "SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" \
              "SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    (cd /tmp && timeout -k 5 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/absq/p$RANDOM -o x -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg2 > /dev/null 2>&1)
  done
  python3 - "$lib" "$KERN" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob("/tmp/absq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if sys.argv[2] in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[1], sys.argv[2], "per launch:", "  ".join(f"{k}={v[0] / v[1]:.4g}" for k, v in sorted(acc.items())))
PY
done
