#!/bin/bash
# Run on the GPU box (through gpurun): the VALU calibration probe bare (wall time + shader clock), then under two PMC
# passes; tools/valu_calibration.py joins them into gpurun_out/TAG_valu_calibration.{txt,json}.
# usage: tools/profile_valu_calib.sh TAG
TAG=${1:-r}
REPO=$PWD
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT
[ -x tools/exp/valu_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/exp/valu_calib.hip -o tools/exp/valu_calib
tools/exp/valu_calib > $OUT/${TAG}_valu_calib_bare.jsonl 2> $OUT/${TAG}_valu_calib.err
cd /tmp
T="timeout -k 5 240"
$T rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_${TAG}_vc1 -o vc1 -- $REPO/tools/exp/valu_calib > $OUT/${TAG}_valu_calib_pmc1.jsonl 2>> $OUT/${TAG}_valu_calib.err
$T rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_${TAG}_vc2 -o vc2 -- $REPO/tools/exp/valu_calib > $OUT/${TAG}_valu_calib_pmc2.jsonl 2>> $OUT/${TAG}_valu_calib.err
cd $REPO
python3 tools/valu_calibration.py $OUT/${TAG}_valu_calibration.txt $OUT/${TAG}_valu_calibration.json $OUT/${TAG}_valu_calib_bare.jsonl $OUT/pmc_${TAG}_vc1 $OUT/pmc_${TAG}_vc2
rm -rf $OUT/pmc_${TAG}_vc1 $OUT/pmc_${TAG}_vc2
