// recurrent.hip — the recurrent-PPO pieces of SURVEY.md 8f.2 (ppo_lstm_lunarlander.py):
//
//   gymrl_gru_cell_fwd / _bwd   the pointwise half of torch.nn.GRU's cell (URNN :449-491).  The two
//                               gate GEMMs (x W_ih^T + b_ih, h W_hh^T + b_hh) are library work; what is
//                               left is 6 reads + 1 write per hidden unit, fused here into one pass
//                               (autograd runs ~12 elementwise launches for the same arithmetic)
//   gymrl_rnd_reward            collect_experience's intrinsic reward  rew += mean((predict - target)^2)
//                               (:588-590), one wave per row
//
// Gate order and formulas are PyTorch's:  r = s(gi_r + gh_r), z = s(gi_z + gh_z),
// n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h, with s(x) = 1 / (1 + exp(-x)) on the
// reproducible det_expf so that the CPU oracle reproduces every bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gymrl.h"
#include "gymrl_device.hpp"

using namespace gymrl;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float det_sigmoidf(float x) { return 1.0f / (1.0f + det_expf(-x)); }

__global__ __launch_bounds__(kBlock) void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                              const float* __restrict__ h, int B, int H,
                                                              float* __restrict__ h_out) {
  const int H4 = H >> 2;
  const int64_t total = (int64_t)B * H4;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t b = e / H4;
    const int c = (int)(e - b * H4) << 2;
    const float* gir = gi + b * 3 * H + c;
    const float* ghr = gh + b * 3 * H + c;
    const float4 ir = *reinterpret_cast<const float4*>(gir), iz = *reinterpret_cast<const float4*>(gir + H),
                 in = *reinterpret_cast<const float4*>(gir + 2 * H);
    const float4 hr = *reinterpret_cast<const float4*>(ghr), hz = *reinterpret_cast<const float4*>(ghr + H),
                 hn = *reinterpret_cast<const float4*>(ghr + 2 * H);
    const float4 hp = *reinterpret_cast<const float4*>(h + b * H + c);
    float4 o;
#define GRU_FWD(x)                                         \
  {                                                        \
    const float r = det_sigmoidf(ir.x + hr.x);             \
    const float z = det_sigmoidf(iz.x + hz.x);             \
    const float n = det_tanhf_sel(in.x + r * hn.x);        \
    o.x = (1.0f - z) * n + z * hp.x;                       \
  }
    GRU_FWD(x) GRU_FWD(y) GRU_FWD(z) GRU_FWD(w)
#undef GRU_FWD
    *reinterpret_cast<float4*>(h_out + b * H + c) = o;
  }
}

// gates are recomputed from (gi, gh): nothing but the cell's inputs has to be kept for backward
__global__ __launch_bounds__(kBlock) void gru_cell_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                              const float* __restrict__ h, const float* __restrict__ dh_out,
                                                              int B, int H, float* __restrict__ dgi,
                                                              float* __restrict__ dgh, float* __restrict__ dh) {
  const int H4 = H >> 2;
  const int64_t total = (int64_t)B * H4;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t b = e / H4;
    const int c = (int)(e - b * H4) << 2;
    const int64_t g0 = b * 3 * H + c;
    const float4 ir = *reinterpret_cast<const float4*>(gi + g0), iz = *reinterpret_cast<const float4*>(gi + g0 + H),
                 in = *reinterpret_cast<const float4*>(gi + g0 + 2 * H);
    const float4 hr = *reinterpret_cast<const float4*>(gh + g0), hz = *reinterpret_cast<const float4*>(gh + g0 + H),
                 hn = *reinterpret_cast<const float4*>(gh + g0 + 2 * H);
    const float4 hp = *reinterpret_cast<const float4*>(h + b * H + c);
    const float4 go = *reinterpret_cast<const float4*>(dh_out + b * H + c);
    float4 dir, diz, din, dhn, dhp;
#define GRU_BWD(x)                                         \
  {                                                        \
    const float r = det_sigmoidf(ir.x + hr.x);             \
    const float z = det_sigmoidf(iz.x + hz.x);             \
    const float n = det_tanhf_sel(in.x + r * hn.x);        \
    const float dn = go.x * (1.0f - z);                    \
    const float dz = go.x * (hp.x - n);                    \
    const float dnp = dn * (1.0f - n * n);                 \
    din.x = dnp;                                           \
    dhn.x = dnp * r;                                       \
    dir.x = (dnp * hn.x) * (r * (1.0f - r));               \
    diz.x = dz * (z * (1.0f - z));                         \
    dhp.x = go.x * z;                                      \
  }
    GRU_BWD(x) GRU_BWD(y) GRU_BWD(z) GRU_BWD(w)
#undef GRU_BWD
    *reinterpret_cast<float4*>(dgi + g0) = dir;
    *reinterpret_cast<float4*>(dgi + g0 + H) = diz;
    *reinterpret_cast<float4*>(dgi + g0 + 2 * H) = din;
    *reinterpret_cast<float4*>(dgh + g0) = dir;
    *reinterpret_cast<float4*>(dgh + g0 + H) = diz;
    *reinterpret_cast<float4*>(dgh + g0 + 2 * H) = dhn;
    *reinterpret_cast<float4*>(dh + b * H + c) = dhp;
  }
}

// one wave per row: lane l sums (p - t)^2 over columns l, l + 64, ... in order, then the shfl_down tree
__global__ __launch_bounds__(kBlock) void rnd_reward_kernel(const float* __restrict__ predict, const float* __restrict__ target,
                                                            int B, int E, float* __restrict__ rew, float* __restrict__ rnd_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* p = predict + (size_t)row * E;
  const float* t = target + (size_t)row * E;
  float s = 0.0f;
  for (int c = lane; c < E; c += 64) { const float d = p[c] - t[c]; s += d * d; }
  s = wave_sumf(s);
  if (lane == 0) {
    const float m = s / (float)E;
    if (rnd_out) rnd_out[row] = m;
    if (rew) rew[row] = rew[row] + m;
  }
}

inline int grid_for(int64_t work) {
  int64_t nb = (work + kBlock - 1) / kBlock;
  return (int)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int gymrl_gru_cell_fwd(const float* gi, const float* gh, const float* h, int B, int H, float* h_out, void* stream) {
  if (!gi || !gh || !h || !h_out || B < 0 || H <= 0 || (H & 3) || !aligned16(gi) || !aligned16(gh) || !aligned16(h) ||
      !aligned16(h_out))
    return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(grid_for((int64_t)B * (H >> 2))), dim3(kBlock), 0, (hipStream_t)stream, gi,
                     gh, h, B, H, h_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_gru_cell_bwd(const float* gi, const float* gh, const float* h, const float* dh_out, int B, int H, float* dgi,
                       float* dgh, float* dh, void* stream) {
  if (!gi || !gh || !h || !dh_out || !dgi || !dgh || !dh || B < 0 || H <= 0 || (H & 3) || !aligned16(gi) ||
      !aligned16(gh) || !aligned16(h) || !aligned16(dh_out) || !aligned16(dgi) || !aligned16(dgh) || !aligned16(dh))
    return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(grid_for((int64_t)B * (H >> 2))), dim3(kBlock), 0, (hipStream_t)stream, gi,
                     gh, h, dh_out, B, H, dgi, dgh, dh);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_rnd_reward(const float* predict, const float* target, int B, int E, float* rew_inout, float* rnd_out,
                     void* stream) {
  if (!predict || !target || (!rew_inout && !rnd_out) || B < 0 || E <= 0) return -22;
  if (B == 0) return 0;
  const int rows_per_block = kBlock / 64;
  hipLaunchKernelGGL(rnd_reward_kernel, dim3((B + rows_per_block - 1) / rows_per_block), dim3(kBlock), 0,
                     (hipStream_t)stream, predict, target, B, E, rew_inout, rnd_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
