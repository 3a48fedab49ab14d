// splatco_amd/csrc/adam.h -- launcher of adam.hip (kept out of common.h: the profile tables under profiles/ are stamped
// with the hash of common.h, and the optimizer pass shares nothing with the kernels they describe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/splatco_raster.h"

namespace scr {
constexpr int ADAM_MAX = 24;      // tensors per launch (the table travels in the kernel arguments: 1.4 KB)
// 0 = launched; 1 = more than 2^31 workgroups in one launch
int launch_adam(int n, const scr_adam_tensor* tensors, double beta1, double beta2, double eps, hipStream_t st);
}  // namespace scr
