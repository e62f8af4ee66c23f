// mlp.hip — whole-network forward of the small actor/critic MLPs in ONE launch (gfx950).
//
// The rollout's policy forward (ppo_lunarlander.py:92-108 get_action/get_value, at N envs per
// vector step) is a chain of 256-wide Linear(+Tanh) layers on a [N, obs] batch that is far too
// small to amortise one library GEMM + one elementwise launch per layer (12 launches, ~90 us
// for N = 4096).  Here a workgroup owns 16 batch rows for the whole network:
//
//   * activations never leave the CU: three [16][256] f32 LDS buffers are ping-ponged
//     between stages;
//   * every Linear is v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate, exact f32): the four
//     waves of the workgroup split the output columns in 16-wide tiles, A (activations)
//     comes from LDS as ds_read_b128 along K, B (weights) comes straight from L2 as one
//     global_load_dwordx4 per lane out of a packed copy laid out as the B-operand register
//     image (gymrl_mlp_pack, refreshed once per parameter update) — all workgroups of an XCD
//     stream the same 0.8 MB of weights, so HBM sees them once;
//   * weight loads are double-buffered in registers in batches of 4 K-blocks x 4 tiles
//     (1 wave per SIMD has the whole 512-VGPR file), bias + activation are applied on the
//     accumulators, and only the network outputs (logits, value) are written to HBM.
//
// K order inside one accumulator: a lane's float4 holds 4 consecutive k, component j feeds
// MFMA j of the K-block, so MFMA j multiplies k = kb + 4*q + j (q = MFMA k index 0..3).  The
// f32 MFMA is an fmaf chain over q, hence the summation order is
//   for kb in 0,16,.. (K rounded up to 64): for j in 0..3: for q in 0..3:
//     acc = fmaf(a[kb+4q+j], w[kb+4q+j], acc)        (zeros beyond in_dim)
// starting from acc = 0, bias added afterwards (the CPU checker restates exactly that order).
#include "mlp_device.hpp"

namespace {

using namespace gymrl;
using namespace gymrl::mlp;

// P[tile][kblock][lane][c] <- W (see load_batch); one thread per packed float4.
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ W, int out_dim, int in_dim,
                                                       float* __restrict__ P) {
  const int nkb = ((in_dim + 63) & ~63) >> 4, ntiles = (out_dim + 15) >> 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntiles * nkb * 64) return;
  const int lane = idx & 63, kbi = (idx >> 6) % nkb, t = (idx >> 6) / nkb;
  const int n = 16 * t + (lane & 15), k = 16 * kbi + 4 * (lane >> 4);
  f32x4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = (n < out_dim && k + c < in_dim) ? W[(size_t)n * in_dim + k + c] : 0.0f;
  reinterpret_cast<f32x4*>(P)[idx] = v;
}

__global__ __launch_bounds__(kWaves * 64) void mlp_forward_kernel(const float* __restrict__ x, int n_rows,
                                                                  int in_dim, gymrl_mlp_desc d) {
  // dynamic LDS, sized by the launcher to more than half of the CU's 160 KB so that exactly one
  // workgroup (4 waves, one per SIMD) is resident per CU: two co-resident workgroups would halve
  // each one's share of the matrix pipe while other CUs idle.
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  float (*lds)[kRows * kStride] = reinterpret_cast<float (*)[kRows * kStride]>(dyn_lds);
  float* xin = dyn_lds + kBufs * kRows * kStride;
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * kRows;
  // input tile -> xin, zero-padded to a multiple of 64 columns (the packed K) and to 16 rows
  {
    const int k16 = (in_dim + 63) & ~63;
    for (int e = tid; e < kRows * k16; e += kWaves * 64) {
      const int r = e / k16, c = e - r * k16;
      float v = 0.0f;
      if (m0 + r < n_rows && c < in_dim) v = x[(size_t)(m0 + r) * in_dim + c];
      xin[r * kInStride + c] = v;
    }
  }
  __syncthreads();
  forward_tile(d, lds, xin, nullptr, m0, n_rows, tid);
}

}  // namespace

extern "C" {

int gymrl_mlp_forward(const float* x, int n_rows, int in_dim, const gymrl_mlp_desc* desc, void* stream) {
  if (!x || !desc || n_rows < 0 || in_dim <= 0 || in_dim > kMaxIn) return -22;
  if (desc->n_stages <= 0 || desc->n_stages > GYMRL_MLP_MAX_STAGES) return -22;
  for (int s = 0; s < desc->n_stages; ++s) {
    const gymrl_mlp_stage& st = desc->stage[s];
    if (!st.W || st.in_dim <= 0 || st.out_dim <= 0 || st.in_dim > kMaxW) return -22;
    if (st.src < -1 || st.src > 2 || st.dst < -1 || st.dst > 2 || (st.dst >= 0 && st.dst == st.src)) return -22;
    if (st.dst >= 0 && st.out_dim > kMaxW) return -22;
    if (st.dst < 0 && (!st.out || st.out_stride < st.out_dim)) return -22;
    if (st.src < 0 && st.in_dim != in_dim) return -22;
    if (st.act < GYMRL_ACT_NONE || st.act > GYMRL_ACT_RELU) return -22;
    if (reinterpret_cast<uintptr_t>(st.W) & 15) return -22;
  }
  if (n_rows == 0) return 0;
  const int blocks = (n_rows + kRows - 1) / kRows;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(mlp_forward_kernel, dim3(blocks), dim3(kWaves * 64), kLdsBytes, (hipStream_t)stream, x, n_rows,
                     in_dim, *desc);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

size_t gymrl_mlp_packed_floats(int out_dim, int in_dim) {
  if (out_dim <= 0 || in_dim <= 0) return 0;
  return (size_t)((out_dim + 15) >> 4) * (size_t)(((in_dim + 63) & ~63) >> 4) * 256;
}

int gymrl_mlp_pack(const float* W, int out_dim, int in_dim, float* packed, void* stream) {
  if (!W || !packed || out_dim <= 0 || in_dim <= 0 || (reinterpret_cast<uintptr_t>(packed) & 15)) return -22;
  const size_t vec = gymrl_mlp_packed_floats(out_dim, in_dim) / 4;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)((vec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, out_dim,
                     in_dim, packed);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
