#!/bin/bash
# Developer A/B of named kernels inside a configuration under rocprofv3 --kernel-trace --stats:
# tools/ab_kern.sh CONFIG "regex" lib [lib ...]   ("base" = product library) -> average microseconds per launch of the kernels whose name matches
CFG=$1; PAT=$2; shift 2
export TMPDIR=/tmp
R=$PWD
for lib in "$@"; do
  if [ "$lib" = "base" ]; then unset SPLATCO_RASTER_LIB; else export SPLATCO_RASTER_LIB=$R/$lib; fi
  D=/tmp/abk_$$
  rm -rf $D
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python3 $R/bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline --no-cfg2 > $D.log 2>&1)
  python3 - "$CFG" "$(basename $lib)" "$PAT" $D <<'PY'
import csv, glob, re, sys
cfg, lib, pat, d = sys.argv[1:5]
out = []
for path in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("scr::", "")
        if re.search(pat, name):
            out.append(f"{name} {float(r['AverageNs']) / 1e3:.1f}us x{r['Calls']}")
print(cfg, lib + ":", " | ".join(sorted(out)))
PY
  rm -rf $D $D.log
done
