// train_device.hpp — device helpers shared by the update-path kernels (mlp_train.hip, gemm.hip).
#pragma once
#include "gymrl_device.hpp"

namespace gymrl {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// tanh for the training forward: |x| < 0.625 -> the odd polynomial of det_tanhf (full
// relative accuracy near 0), else 1 - 2/(exp(2x)+1) on the hardware exp2 / rcp units
// (~2 ulp).  Deterministic on the device, not restated on the CPU: the update's forward is
// compared with torch at 1e-5, not bit for bit.
__device__ __forceinline__ float fast_tanhf(float x) {
  const float a = __builtin_fabsf(x);
  const float e = __builtin_amdgcn_exp2f(__builtin_fminf(a, 10.0f) * 2.885390081777927f);  // exp(2a)
  float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  big = x < 0.0f ? -big : big;
  const float z = x * x;
  float p = -5.70498872745e-3f;
  p = fmaf(p, z, 2.06390887954e-2f);
  p = fmaf(p, z, -5.37397155531e-2f);
  p = fmaf(p, z, 1.33314422036e-1f);
  p = fmaf(p, z, -3.33332819422e-1f);
  const float small = fmaf(p * z, x, x);
  return a >= 0.625f ? big : small;
}

}  // namespace gymrl
