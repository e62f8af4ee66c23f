// gemm_sb.hip — the update's 256-wide GEMMs on the bf16 matrix cores with f32-accurate products ("split-bf16").
//
// gfx950 runs `v_mfma_f32_32x32x16_bf16` at 16 x the rate of the exact-f32 `v_mfma_f32_32x32x2_f32` that
// csrc/gemm.hip is built on (2.5 PFLOP/s against 157 TFLOP/s dense).  A float splits EXACTLY into three
// bf16-representable pieces by truncation — hi = the top 8 significant bits, mid = the next 8, lo = the last 8
// (x - hi and x - hi - mid are exact in f32) — and a product of two bf16 values is exact in the matrix core's f32
// accumulator, so
//
//     a * b = (ah + am + al)(bh + bm + bl) = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + O(2^-24 |a b|)
//
// six bf16 MFMAs reproduce an f32 product to ~1.2e-7 relative (the three dropped terms), accumulated in f32 like
// any f32 GEMM: 6/16 of the exact path's matrix time.  This is an f32-ACCURATE GEMM (measured against float64 next
// to the exact kernels and torch's own f32 GEMM in tests/test_gemm_sb_gpu.py), not a bit-exact restatement of an
// fmaf chain: the exact kernels stay the default and the ones the CPU oracle pins; this path is opt-in
// (a probe for now: see below).
//
// Forward / input gradient: weight-stationary like gemm.hip — a workgroup keeps a [256 x 64] slice of the weights in
// LDS for its whole life, already split into three bf16 planes laid out as the MFMA operand image ([plane][k / 8]
// [n][8 k]: one ds_read_b128 per plane, tile and 16-deep step), and streams 64-row tiles of the activations past it;
// a lane loads the 32 bytes X[row][16 s + 8 g .. + 7] of its row straight into the operand layout and splits them on
// the VALU, which issues in the gaps of the 32-cycle MFMAs (<= 5 slots per gap with one wave per SIMD).  Operands
// are swapped (the instruction's A = weights, B = activations), so a lane's accumulator quads are four adjacent
// output columns of one row: 16-byte stores with the bias / tanh epilogue on them.
// PROBE, not part of libgymrl_hip.so: tools/micro_gemm_sb.py builds it (hipcc -shared -I gymrl_amd/csrc -I include) and
// reports what it found — products accurate to 5.3e-7 of max|y| against float64 (the exact fmaf-chain kernel: 6.9e-7, the
// library GEMM: 7.6e-7); matrix + split + store work 112 us per 262144 x 256 x 256 launch; but with the operand loads as
// written here (one row per lane, 32 bytes per step) the load stream and the compute do not overlap: 280 us, no faster
// than the exact kernel (289).  What it needs is the A tile staged through LDS by coalesced direct-to-LDS loads.
#include "train_device.hpp"
#include "gymrl.h"

#ifndef SB_ABL
#define SB_ABL 0          // 1: no operand split, 2: no MFMAs (timing ablations of tools/micro_gemm_sb.py; wrong results)
#endif

namespace {

using namespace gymrl;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int kBN = 64;          // output columns per workgroup (LDS: 3 planes x 256 x 64 x 2 B = 96 KiB)
constexpr int kRED = 256;        // reduction length held in LDS
constexpr int kThreads = 512;
constexpr int kLdsBytes = 3 * (kRED / 8) * kBN * 16;

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v, v), r, off, 0, 0);
}

// x = hi + mid + lo exactly; each piece has <= 8 significant bits (its f32 encoding's low half is zero)
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, h);
  m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, m);
  l = __builtin_bit_cast(uint32_t, r2);
}
// two f32 encodings with zero low halves -> one register of two bf16 (element 0 in the low half)
__device__ __forceinline__ uint32_t pack2(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

// 8 consecutive floats -> the three bf16x8 operand registers
__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, u32x4& H, u32x4& M, u32x4& L) {
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { split3(lo4[e], h[e], m[e], l[e]); split3(hi4[e], h[4 + e], m[4 + e], l[4 + e]); }
#pragma unroll
  for (int e = 0; e < 4; ++e) { H[e] = pack2(h[2 * e], h[2 * e + 1]); M[e] = pack2(m[2 * e], m[2 * e + 1]); L[e] = pack2(l[2 * e], l[2 * e + 1]); }
}

struct SbArgs {
  const float* A; int64_t M; int lda;
  const float* W; int ldw;
  float* out; int ldo;
  const float* bias;
  int act, slices, rows_per_group;
};

// six products, smallest first
__device__ __forceinline__ f32x16 mac6(const u32x4& wh, const u32x4& wm, const u32x4& wl, const u32x4& xh, const u32x4& xm,
                                       const u32x4& xl, f32x16 acc) {
  acc = mfma_bf16(wm, xm, acc);
  acc = mfma_bf16(wl, xh, acc);
  acc = mfma_bf16(wh, xl, acc);
  acc = mfma_bf16(wm, xh, acc);
  acc = mfma_bf16(wh, xm, acc);
  acc = mfma_bf16(wh, xh, acc);
  return acc;
}

__global__ __launch_bounds__(kThreads) void gemm_sb_fwd_kernel(const SbArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];      // [plane][kg][n][4 dwords]
  u32x4* ldsv = reinterpret_cast<u32x4*>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, g = lane >> 5;
  // blocks b, b + 8, ... run on the same XCD: the column slices of one row group are neighbours there (L2 hits)
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int slice = rest % a.slices, grp = (rest / a.slices) * 8 + xcd;
  const int n0 = slice * kBN;
  for (int idx = threadIdx.x; idx < kBN * (kRED / 8); idx += kThreads) {
    const int n = idx / (kRED / 8), kg = idx % (kRED / 8);
    const float* src = a.W + (size_t)(n0 + n) * a.ldw + 8 * kg;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + 4);
    u32x4 H, M, L;
    split8(w0, w1, H, M, L);
    ldsv[(0 * (kRED / 8) + kg) * kBN + n] = H;
    ldsv[(1 * (kRED / 8) + kg) * kBN + n] = M;
    ldsv[(2 * (kRED / 8) + kg) * kBN + n] = L;
  }
  __syncthreads();
  const int64_t row_lo = (int64_t)grp * a.rows_per_group;
  const int64_t row_hi = row_lo + a.rows_per_group < a.M ? row_lo + a.rows_per_group : a.M;
  const __amdgpu_buffer_rsrc_t rA = rsrc_of(a.A, (uint32_t)(a.M * a.lda * 4));
  const __amdgpu_buffer_rsrc_t rO = rsrc_of(a.out, (uint32_t)(a.M * a.ldo * 4));
  const float* bias = a.bias;
  // A ring of kRing 16-deep steps is in flight ahead of the MFMAs (a step is >= 768 matrix cycles; HBM latency ~2 us),
  // and the first steps of the NEXT row tile are issued before the epilogue of the current one.
  constexpr int kRing = 4, kSteps = kRED / 16;
  f32x4 x[kRing][2][2];                                    // [slot][rt][half]
  uint32_t off[2];
  auto offsets = [&](int64_t base_) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t row = base_ + 32 * rt + j;
      off[rt] = row < row_hi ? (uint32_t)((row * a.lda + 8 * g) * 4) : 0x80000000u;     // past the end: reads 0
#if SB_ABL == 5   // (timing only) the same bytes per instruction from 4 rows x 256 contiguous bytes instead of 32 rows x 2 x 16
      off[rt] = (uint32_t)(((base_ + 32 * rt + (lane >> 4)) * a.lda) * 4 + (lane & 15) * 16);
#endif
    }
  };
  auto issue = [&](int slot, int step) {
#if SB_ABL == 4
    if (step >= 0) return;
#endif
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      x[slot][rt][0] = ld4(rA, off[rt] + 64 * step);
      x[slot][rt][1] = ld4(rA, off[rt] + 64 * step + 16);
    }
  };
  const int64_t stride = 64 * (kThreads / 64);
  int64_t base = row_lo + 64 * wave;
  if (base < row_hi) {
    offsets(base);
#pragma unroll
    for (int s = 0; s < kRing; ++s) issue(s, s);
  }
  for (; base < row_hi; base += stride) {
    f32x16 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][nt][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const int slot = s % kRing;
      u32x4 xh[2], xm[2], xl[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#if SB_ABL == 1
        xh[rt] = __builtin_bit_cast(u32x4, x[slot][rt][0]); xm[rt] = __builtin_bit_cast(u32x4, x[slot][rt][1]); xl[rt] = xh[rt];
#else
        split8(x[slot][rt][0], x[slot][rt][1], xh[rt], xm[rt], xl[rt]);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + kRing < kSteps) {
        issue(slot, s + kRing);
      } else if (base + stride < row_hi) {                 // this slot is free: start the next tile
        if (s + kRing == kSteps) offsets(base + stride);
        issue(slot, s + kRing - kSteps);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int kg = 2 * s + g, n = 32 * nt + j;
        const u32x4 wh = ldsv[(0 * (kRED / 8) + kg) * kBN + n];
        const u32x4 wm = ldsv[(1 * (kRED / 8) + kg) * kBN + n];
        const u32x4 wl = ldsv[(2 * (kRED / 8) + kg) * kBN + n];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#if SB_ABL == 2
          acc[rt][nt][0] += __builtin_bit_cast(float, wh[0] ^ wm[1] ^ wl[2] ^ xh[rt][0] ^ xm[rt][1] ^ xl[rt][2]);
#else
          acc[rt][nt] = mac6(wh, wm, wl, xh[rt], xm[rt], xl[rt], acc[rt][nt]);
#endif
        }
      }
    }
    // epilogue: lane (j, g) holds row base + 32 rt + j, columns n0 + 32 nt + 8 q + 4 g .. + 3 in accumulator quad q
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t row = base + 32 * rt + j;
      const uint32_t obase = row < row_hi ? (uint32_t)((row * a.ldo + n0 + 4 * g) * 4) : 0x80000000u;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + 32 * nt + 8 * q + 4 * g;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float z = acc[rt][nt][4 * q + e] + (bias ? bias[n + e] : 0.0f);
            v[e] = a.act ? train_tanhf(z) : z;
          }
#if SB_ABL == 3
          if (v[0] == 123.456f) st4(rO, obase + (32 * nt + 8 * q) * 4, v);
#else
          st4(rO, obase + (32 * nt + 8 * q) * 4, v);
#endif
        }
    }
  }
}

}  // namespace

extern "C" {

// Y [B, N] = act(X [B, 256] W[N, 256]^T + b): f32-accurate products on the bf16 matrix cores (see the header)
int gymrl_linear_fwd_split_bf16(const float* X, const float* W, const float* b, int64_t B, int K, int N, int act, float* Y,
                                void* stream) {
  if (!X || !W || !Y || B < 0 || K != kRED || N < kBN || N % kBN || N > 512) return -22;
  if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15) return -22;
  if (B == 0) return 0;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_sb_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr = true;
  }
  SbArgs a{};
  a.A = X; a.M = B; a.lda = K; a.W = W; a.ldw = K; a.out = Y; a.ldo = N; a.bias = b; a.act = act;
  a.slices = N / kBN;
  int groups = 256 / a.slices;                                  // one workgroup per CU
  groups = (groups / 8) * 8;
  if (groups < 8) groups = 8;
  int64_t per = (B + groups - 1) / groups;
  per = (per + 63) / 64 * 64;
  a.rows_per_group = (int)per;
  hipLaunchKernelGGL(gemm_sb_fwd_kernel, dim3(groups * a.slices), dim3(kThreads), kLdsBytes, (hipStream_t)stream, a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
