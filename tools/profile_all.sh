#!/bin/bash
# Run on the GPU box (through gpurun): the round's profile set, reduced to the text summaries that go under profiles/.
# usage: tools/profile_all.sh TAG   -> gpurun_out/TAG_kernel_stats.txt, TAG_cfg2_kernel_stats.txt, TAG_pmc_summary.txt,
#        hbm_traffic_TAG.json; the raw rocprofv3 output (hundreds of MB) is deleted on the box.
TAG=${1:-r}
OUT=$PWD/gpurun_out
bash tools/profile_round.sh $TAG > $OUT/profile_$TAG.log 2>&1
DB=$(find $OUT/prof_$TAG -name "*.db" | head -1)
python3 tools/rocpd_summary.py $DB $OUT/${TAG}_kernel_stats.txt "bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg2 (rocprofv3 --kernel-trace --stats)" > /dev/null
python3 tools/pmc_summary.py $OUT/${TAG}_pmc_summary.txt $OUT/hbm_traffic_$TAG.json $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/pmc_${TAG}_sq1 $OUT/pmc_${TAG}_sq2 > /dev/null
bash tools/profile_cfg.sh ${TAG}_cfg2 cfg2 --steps 30 > $OUT/profile_${TAG}_cfg2.log 2>&1
rm -rf $OUT/prof_$TAG $OUT/prof_${TAG}_cfg2 $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/pmc_${TAG}_sq1 $OUT/pmc_${TAG}_sq2
head -12 $OUT/${TAG}_kernel_stats.txt; head -20 $OUT/${TAG}_pmc_summary.txt | cut -c1-220; du -sh $OUT
