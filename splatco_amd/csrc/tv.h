// splatco_amd/csrc/tv.h -- launcher of tv.hip (kept out of common.h for the same reason as adam.h: the profile tables
// under profiles/ are stamped with the hash of common.h, and this pass shares nothing with the kernels they describe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/splatco_raster.h"

namespace scr {
constexpr int TV_MAX = 12;        // planes per launch (three projections of up to four grids)
// 0 = launched (or nothing to do); 1 = more than 2^31 workgroups in one launch
int launch_tv_add_grad(int n, const scr_tv_plane* planes, hipStream_t st);
}  // namespace scr
