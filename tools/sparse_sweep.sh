#!/bin/bash
# Run on the GPU box (through gpurun): bench.py cfg1 at sigma scales 1, 0.5, 0.25, 0.1 (same Gaussians, sparser tile lists),
# kernel-trace stats of every point and the PMC passes of the sparsest.  usage: tools/sparse_sweep.sh TAG
TAG=${1:-r04_sparse}
REPO=$PWD
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
for S in 1 0.5 0.25 0.1; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 --sigma-scale $S > $OUT/${TAG}_s${S}.json 2> $OUT/${TAG}_s${S}.err
  (cd /tmp && timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_s${S} -o stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 --sigma-scale $S > $OUT/${TAG}_s${S}_prof.log 2>&1)
  DB=$(find $OUT/prof_${TAG}_s${S} -name "*.db" | head -1)
  python3 tools/rocpd_summary.py $DB $OUT/${TAG}_s${S}_kernel_stats.txt "bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 --sigma-scale $S (rocprofv3 --kernel-trace --stats)" > /dev/null
  rm -rf $OUT/prof_${TAG}_s${S}
done
BENCH_ARGS="--sigma-scale 0.1" bash tools/profile_round.sh ${TAG}_s0.1 pmc-only > $OUT/${TAG}_pmc.log 2>&1
python3 tools/pmc_summary.py $OUT/${TAG}_s0.1_pmc_summary.txt $OUT/hbm_traffic_${TAG}_s0.1.json $OUT/pmc_${TAG}_s0.1_fetch $OUT/pmc_${TAG}_s0.1_write $OUT/pmc_${TAG}_s0.1_sq1 $OUT/pmc_${TAG}_s0.1_sq2 > /dev/null
rm -rf $OUT/pmc_${TAG}_s0.1_fetch $OUT/pmc_${TAG}_s0.1_write $OUT/pmc_${TAG}_s0.1_sq1 $OUT/pmc_${TAG}_s0.1_sq2
for S in 1 0.5 0.25 0.1; do python3 - $OUT/${TAG}_s${S}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c, k = d["config"], d["kernel_rooflines"]
print(f"sigma x {c.get('sigma_scale', 1.0)}: I = {c['tile_instances']} ({c.get('mean_list_entries_per_tile', c['tile_instances'] / 8160):.0f} per tile), step {d['ms_per_step']:.3f} ms, {d['value']:.0f} Msplats/s; "
      + "; ".join(f"{n.replace('_kernel', '')} {k[n]['ms'] * 1e3:.0f} us {k[n]['GBps']:.0f} GB/s ({k[n]['frac_of_8TBps']:.3f} of 8 TB/s, {k[n]['frac_of_measured_peak']:.3f} of the copy peak)"
                  for n in ("blend_forward_kernel", "blend_backward_kernel")))
PY
done
head -30 $OUT/${TAG}_s0.1_pmc_summary.txt | cut -c1-250
