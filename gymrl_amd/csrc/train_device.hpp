// train_device.hpp — device helpers shared by the update-path kernels (mlp_train.hip, gemm.hip).
#pragma once
#include "gymrl_device.hpp"

namespace gymrl {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// tanh of the training forward: 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units — 5 VALU instructions,
// exact limits (+-1) for large |x|, absolute error < 2e-7 everywhere (relative accuracy is lost only where
// |tanh| < 1e-3).  The f32 MFMA shares the SIMD's f32 lanes with the VALU, so an activation in a GEMM epilogue is
// paid in matrix time: the odd-polynomial form used in round 1 (20 instructions) cost 25 % of the kernel.
// Deterministic on the device, not restated on the CPU: the update's forward is compared with torch at 1e-5.
__device__ __forceinline__ float train_tanhf(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);                         // exp(2x)
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

}  // namespace gymrl
