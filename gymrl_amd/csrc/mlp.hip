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
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

namespace {

using namespace gymrl;

constexpr int kRows   = 16;                 // batch rows per workgroup = one MFMA M tile
constexpr int kMaxW   = GYMRL_MLP_MAX_WIDTH; // widest layer input/output held in LDS
constexpr int kStride = kMaxW + 4;          // +4 floats: rows start on different banks
constexpr int kBufs   = 3;                  // ping-pong activation buffers
constexpr int kMaxIn  = GYMRL_MLP_MAX_INPUT; // widest network input (observation)
constexpr int kInStride = kMaxIn + 4;   // kMaxIn is a multiple of 64
constexpr int kWaves  = 4;
constexpr int kLdsBytes = 96 * 1024;        // > 80 KB: one workgroup per CU
constexpr int kTPW    = 4;                  // 16-column tiles a wave accumulates together

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float activate(float x, int act) {
  if (act == GYMRL_ACT_TANH) return det_tanhf_sel(x);
  if (act == GYMRL_ACT_RELU) return fmaxf(x, 0.0f);
  return x;
}

// Register batch of weights: UK K-blocks (16 k each) x NT tiles, one float4 per lane.
template <int NT, int UK>
struct Batch { f32x4 w[UK][NT]; };

// Weights are read from the PACKED layout gymrl_mlp_pack writes (once per parameter update):
//   P[tile][kblock][lane][c] = W[16*tile + (lane&15)][16*kblock + 4*(lane>>4) + c]
// zero-padded to whole tiles and to K rounded up to 64, i.e. exactly the MFMA B-operand
// register image.  A wave-wide dwordx4 load is then ONE contiguous 1-KiB segment (8 full
// cache lines, consecutive loads walk consecutive L2 channels) and needs no bounds checks.
// Reading torch's [out][in] rows directly costs 16 half-lines per load whose 1-KiB row
// stride lands on 4 of the 16 L2 channels: 8.4 us per 256x256 layer instead of < 2.
template <int NT, int UK>
__device__ __forceinline__ void load_batch(Batch<NT, UK>& b, const float* __restrict__ P, int nkb, int kb0, int t0,
                                           int lane) {
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const f32x4* tile = reinterpret_cast<const f32x4*>(P + ((size_t)(t0 + i) * nkb + kb0) * 256) + lane;
#pragma unroll
    for (int u = 0; u < UK; ++u) b.w[u][i] = tile[64 * u];
  }
}

template <int NT, int UK>
__device__ __forceinline__ void mfma_batch(f32x4 (&acc)[NT], const Batch<NT, UK>& b, const float* __restrict__ A,
                                           int a_stride, int kb0, int lane) {
  const float* ap = A + (lane & 15) * a_stride + 16 * kb0 + 4 * (lane >> 4);
#pragma unroll
  for (int u = 0; u < UK; ++u) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 16 * u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b.w[u][i][j], acc[i], 0, 0, 0);
    }
  }
}

// NT adjacent 16-column tiles of one stage, K = NB batches of UK K-blocks: straight-line code, the
// next batch's loads are issued before the current batch's MFMAs, so the compiler's vmcnt waits
// leave exactly one batch in flight behind the matrix pipe.
template <int NT, int UK, int NB>
__device__ __forceinline__ void run_tiles_k(const gymrl_mlp_stage& st, const float* __restrict__ A, int a_stride,
                                            float* __restrict__ dst_lds, int t0, int m0, int n_rows, int lane) {
  const int nkb = ((st.in_dim + 63) & ~63) >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  Batch<NT, UK> b[2];
  load_batch<NT, UK>(b[0], st.W, nkb, 0, t0, lane);
  float bias[NT];                      // issued now, consumed in the epilogue
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = min(16 * (t0 + i) + (lane & 15), st.out_dim - 1);
    bias[i] = st.b ? st.b[col] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (i + 1 < NB) load_batch<NT, UK>(b[(i + 1) & 1], st.W, nkb, (i + 1) * UK, t0, lane);
    // keep the machine scheduler from sinking the prefetch down to its first use
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch<NT, UK>(acc, b[i & 1], A, a_stride, i * UK, lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue: D[row = 4*(lane>>4)+r][col = lane&15] + bias, activation, to LDS or HBM
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = 16 * (t0 + i) + (lane & 15);
    if (col < st.out_dim) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        const float v = activate(acc[i][r] + bias[i], st.act);
        if (dst_lds) dst_lds[row * kStride + col] = v;
        else if (m0 + row < n_rows) st.out[(size_t)(m0 + row) * st.out_stride + col] = v;
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void run_tiles(const gymrl_mlp_stage& st, const float* A, int a_stride, float* dst_lds,
                                          int t0, int m0, int n_rows, int lane) {
  const int nb64 = (st.in_dim + 63) >> 6;      // batches of 4 K-blocks; <= 4 for widths <= 256
#define GYMRL_MLP_CASE(UK_, NB_) run_tiles_k<NT, UK_, NB_>(st, A, a_stride, dst_lds, t0, m0, n_rows, lane)
  // up to 16 float4 weight loads per lane in flight per batch
  if (nb64 == 4)      { if constexpr (NT == 1) GYMRL_MLP_CASE(16, 1); else if constexpr (NT == 2) GYMRL_MLP_CASE(8, 2); else GYMRL_MLP_CASE(4, 4); }
  else if (nb64 == 2) { if constexpr (NT <= 2) GYMRL_MLP_CASE(8, 1); else GYMRL_MLP_CASE(4, 2); }
  else if (nb64 == 1) GYMRL_MLP_CASE(4, 1);
  else GYMRL_MLP_CASE(4, 3);
#undef GYMRL_MLP_CASE
}

__device__ __forceinline__ void run_stage(const gymrl_mlp_stage& st, const float* A, int a_stride, float* dst_lds,
                                          int m0, int n_rows, int lane, int wave) {
  const int ntiles = (st.out_dim + 15) >> 4;
  const int per_wave = (ntiles + kWaves - 1) / kWaves;
  for (int tb = 0; tb < per_wave; tb += kTPW) {              // one pass for widths <= 256
    const int t0 = wave * per_wave + tb;
    const int ntile = min(kTPW, min(per_wave - tb, ntiles - t0));
    if (ntile >= 4) run_tiles<4>(st, A, a_stride, dst_lds, t0, m0, n_rows, lane);
    else if (ntile == 3) run_tiles<3>(st, A, a_stride, dst_lds, t0, m0, n_rows, lane);
    else if (ntile == 2) run_tiles<2>(st, A, a_stride, dst_lds, t0, m0, n_rows, lane);
    else if (ntile == 1) run_tiles<1>(st, A, a_stride, dst_lds, t0, m0, n_rows, lane);
  }
}

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
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  for (int s = 0; s < d.n_stages; ++s) {
    const gymrl_mlp_stage st = d.stage[s];
    const float* A = st.src < 0 ? xin : lds[st.src];
    const int a_stride = st.src < 0 ? kInStride : kStride;
    float* dst_lds = st.dst >= 0 ? lds[st.dst] : nullptr;
    run_stage(st, A, a_stride, dst_lds, m0, n_rows, lane, wave);
    // columns between out_dim and the next multiple of 64 must read as zero: they are the next stage's K padding
    if (st.dst >= 0 && (st.out_dim & 63)) {
      const int lo = st.out_dim, hi = (st.out_dim + 63) & ~63;
      for (int e = tid; e < kRows * (hi - lo); e += kWaves * 64) {
        const int r = e / (hi - lo), c = lo + e % (hi - lo);
        lds[st.dst][r * kStride + c] = 0.0f;
      }
    }
    __syncthreads();
  }
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
