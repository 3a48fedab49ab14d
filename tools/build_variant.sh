#!/bin/bash
# Developer A/B: a variant library with ONE source rebuilt under extra flags.
# usage: tools/build_variant.sh NAME file.hip [-DFLAG ...]  -> splatco_amd/csrc/exp/libvar_NAME.so (use with SPLATCO_RASTER_LIB / tools/ab_kern.sh)
NAME=$1; SRC=$2; shift 2
cd $(dirname $0)/../splatco_amd/csrc
mkdir -p exp
EXTRA=""
[ "$SRC" = "blend.hip" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $EXTRA "$@" -c $SRC -o exp/${SRC%.hip}_$NAME.o || exit 1
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS exp/${SRC%.hip}_$NAME.o -o exp/libvar_$NAME.so && echo exp/libvar_$NAME.so
