#!/bin/bash
# Run on the GPU box (through gpurun): the round's final profile set on the frozen kernels.
# usage: SPLATCO_GIT_SHA=<sha> tools/profile_final.sh TAG
#   cfg1: kernel stats + PMC tables (hbm_traffic / valu_insts / valu_busy / gui_active)      [tools/profile_all.sh]
#   cfg2: kernel stats (profile_all) + HBM traffic table; cfg3 / cfg4 (one rank): bench line + kernel stats; cfg4 HBM traffic
TAG=${1:-r}
OUT=$PWD/gpurun_out
mkdir -p $OUT
bash tools/profile_all.sh $TAG > $OUT/profile_final_$TAG.log 2>&1
bash tools/profile_cfg_pmc.sh $TAG cfg2 >> $OUT/profile_final_$TAG.log 2>&1
for CFG in cfg3 cfg4; do
  bash tools/profile_cfg.sh ${TAG}_$CFG $CFG --steps 8 --warmup 3 >> $OUT/profile_final_$TAG.log 2>&1
  grep '^{"metric"' $OUT/bench_${TAG}_$CFG.log | tail -1 > $OUT/${TAG}_bench_$CFG.json
  rm -rf $OUT/prof_${TAG}_$CFG
done
bash tools/profile_cfg_pmc.sh $TAG cfg4 >> $OUT/profile_final_$TAG.log 2>&1
python3 bench.py > $OUT/${TAG}_bench_cfg1.json 2> $OUT/${TAG}_bench_cfg1.err
ls -la $OUT | grep $TAG | head -40
tail -5 $OUT/profile_final_$TAG.log
