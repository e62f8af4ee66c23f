// mlp_train.hip — the HBM-bound pieces of the ActorCritic update around the library GEMMs.
//
// One PPO minibatch (ppo_lunarlander.py:274-307) pushes B = T*N/32 rows (262,144 at
// BASELINE config 2) through six 256-wide layers.  The 256x256 contractions are MFMA-bound
// library GEMMs at ~90 % of the f32 matrix peak; everything else autograd launches around
// them is one full pass over a [B, 256] activation per op (tanh, tanh', bias-gradient
// reductions, gradient accumulation of the two heads, K = 8 / N = 4 "GEMMs" that only move
// memory) — 36 % of the update's GPU time.  These kernels fuse those passes:
//
//   linear_tanh_smallk      H1 = tanh(X W1^T + b1), K = obs_dim <= 16: VALU dot products, one
//                           write of H1 (replaces a K=8 GEMM tile-padded to 32 + a tanh pass)
//   tanh_inplace            H = tanh(Z) after a library GEMM
//   heads_bwd               both heads' backward in ONE pass over [Ha | Hc]: dZ = (dOut W) * (1 - H^2)
//                           written once, plus dW2 / db2 of both heads and the bias gradient
//                           of the layer below (replaces 4 skinny GEMMs, 2 tanh', 4 reductions
//                           and the accumulation of the two heads' trunk gradients)
//   tanh_bwd_colsum         dZ = dH * (1 - H^2) in place + column sums (bias gradient)
//   linear_smallk_bwd       first layer: dZ1 = dH1 * (1 - H1^2) is never written; dW1, db1
//                           are accumulated straight from dH1, H1, X
//
// Row-major [B, C] activations, C a power of two <= 256; a row is C/4 adjacent lanes x float4
// so a wavefront touches 64/(C/4) whole rows per load.  Column reductions are deterministic:
// per-lane serial f32 sums over the rows a lane visits, LDS tree inside the workgroup, one f32
// partial row per workgroup, fixed-order f64 finalize.
#include "train_device.hpp"
#include "finalize_device.hpp"
#include "policy_device.hpp"
#include "../../include/gymrl.h"

namespace {

using namespace gymrl;

constexpr int kTB = 256;        // threads per workgroup (4 waves)
constexpr int kMaxBlocks = 1024;

struct RowMap {            // lane -> (row within the wave's group, float4 column)
  int lpr, rpw, r_in, c4;
  __device__ RowMap(int C, int lane) : lpr(C >> 2), rpw(64 / (C >> 2)), r_in(lane / (C >> 2)), c4(lane % (C >> 2)) {}
};

inline int grid_for(int64_t rows, int C) {
  const int rows_per_block = (kTB / 64) * (64 / (C / 4));
  int64_t nb = (rows + rows_per_block - 1) / rows_per_block;
  return (int)(nb < kMaxBlocks ? (nb < 1 ? 1 : nb) : kMaxBlocks);
}

// Workgroup reduction of NV per-lane column accumulators (each lane owns 4 adjacent columns
// x NV/4 logical vectors) over the rpw row groups of each wave and the 4 waves; the result
// row (NV/4 vectors of C columns) goes to partials[blockIdx][v*C + col].
template <int NV>
__device__ __forceinline__ void block_colsum(const float (&acc)[NV], int C, const RowMap& m, float* sm,
                                             float* __restrict__ partials) {
  static_assert(NV % 4 == 0, "vectors of 4 columns");
  const int wave = threadIdx.x >> 6;
  const int groups = (kTB / 64) * m.rpw;           // row groups in the workgroup
  const int g = wave * m.rpw + m.r_in;
  const int width = (NV / 4) * C;
  // sm[g][v*C + 4*c4 + j]
#pragma unroll
  for (int v = 0; v < NV / 4; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) sm[(size_t)g * width + v * C + 4 * m.c4 + j] = acc[4 * v + j];
  __syncthreads();
  for (int e = threadIdx.x; e < width; e += kTB) {
    float s = 0.0f;
    for (int gg = 0; gg < groups; ++gg) s += sm[(size_t)gg * width + e];
    partials[(size_t)blockIdx.x * width + e] = s;
  }
}

// Sum of partials[b][e] over b for one output element per 8 threads: thread (te = tid % 32,
// tb = tid / 32) accumulates blocks tb, tb+8, .. in f64 (independent, pipelined loads), the 8
// sub-sums are combined in tb order through LDS.  Returns the sum to the tb == 0 thread.
constexpr int kFinE = 32, kFinB = kTB / kFinE;
__device__ __forceinline__ double reduce_partials(const float* __restrict__ partials, int nblocks, int width, int e,
                                                  bool valid) {
  __shared__ double fin[kFinB][kFinE];
  const int te = threadIdx.x % kFinE, tb = threadIdx.x / kFinE;
  double s = 0.0;
  if (valid) {
    int b = tb;
    for (; b + 3 * kFinB < nblocks; b += 4 * kFinB) {
      const float p0 = partials[(size_t)b * width + e], p1 = partials[(size_t)(b + kFinB) * width + e];
      const float p2 = partials[(size_t)(b + 2 * kFinB) * width + e], p3 = partials[(size_t)(b + 3 * kFinB) * width + e];
      s += (double)p0; s += (double)p1; s += (double)p2; s += (double)p3;
    }
    for (; b < nblocks; b += kFinB) s += (double)partials[(size_t)b * width + e];
  }
  fin[tb][te] = s;
  __syncthreads();
  double t = 0.0;
  if (tb == 0) {
#pragma unroll
    for (int k = 0; k < kFinB; ++k) t += fin[k][te];
  }
  return t;
}

// One input row x[0..D): every lane of a row group reads the same D floats (one request per
// wave); whole float4s when D allows it.
template <int D>
__device__ __forceinline__ void load_row(float (&xv)[D], const float* __restrict__ p) {
  if constexpr (D % 4 == 0) {
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + d);
      xv[d] = v[0]; xv[d + 1] = v[1]; xv[d + 2] = v[2]; xv[d + 3] = v[3];
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) xv[d] = p[d];
  }
}

// ------------------------------------------------------------------ F1 -------
template <int D, bool TANH = true>
__global__ __launch_bounds__(kTB) void linear_tanh_smallk_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                 const float* __restrict__ b, int64_t B, int C,
                                                                 float* __restrict__ out) {
  const RowMap m(C, threadIdx.x & 63);
  float w[4][D], bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bias[j] = b ? b[4 * m.c4 + j] : 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) w[j][d] = W[(size_t)(4 * m.c4 + j) * D + d];
  }
  const int64_t stride = (int64_t)gridDim.x * (kTB / 64) * m.rpw;
  for (int64_t r = ((int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6)) * m.rpw + m.r_in; r < B; r += stride) {
    float xv[D];
    load_row<D>(xv, x + r * D);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) acc = fmaf(xv[d], w[j][d], acc);
      o[j] = TANH ? train_tanhf(acc + bias[j]) : acc + bias[j];
    }
    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(out + r * C + 4 * m.c4));
  }
}

// z <- tanh(z + bias[col]) (bias == nullptr: no bias); C4 = columns / 4, a power of two.
__global__ __launch_bounds__(kTB) void tanh_inplace_kernel(float* __restrict__ z, int64_t n4, const float* __restrict__ bias,
                                                           int C4) {
  f32x4* p = reinterpret_cast<f32x4*>(z);
  for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kTB) {
    f32x4 v = p[i];
    if (bias) {
      const f32x4 b = reinterpret_cast<const f32x4*>(bias)[i & (C4 - 1)];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += b[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = train_tanhf(v[j]);
    p[i] = v;
  }
}

// ------------------------------------------------------------------ F3/F4 ----
// Zac [B, 2C] (pre-activations of actor.0 | critic.0, bias included) -> Hac = tanh(Zac) in place,
// logits[r][a] = ba2[a] + sum_j Ha[r][j] Wa2[a][j], value[r] = bc2 + sum_j Hc[r][j] Wc2[j]: the
// heads are evaluated on the tanh values while they are still in registers, so Hac is not read
// again by two skinny GEMMs.  The dot products are reduced across the C/4 lanes of a row by an
// xor butterfly (fixed order).
template <int A>
__global__ __launch_bounds__(kTB) void heads_fwd_tanh_kernel(float* __restrict__ Zac, int64_t B, int C,
                                                             const float* __restrict__ bac,
                                                             const float* __restrict__ Wa2, const float* __restrict__ ba2,
                                                             const float* __restrict__ Wc2, const float* __restrict__ bc2,
                                                             float* __restrict__ logits, float* __restrict__ value,
                                                             int store_h) {
  const RowMap m(C, threadIdx.x & 63);
  float wa[A][4], wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wc[j] = Wc2[4 * m.c4 + j];
#pragma unroll
    for (int a = 0; a < A; ++a) wa[a][j] = Wa2[(size_t)a * C + 4 * m.c4 + j];
  }
  float bias[A + 1];
#pragma unroll
  for (int a = 0; a < A; ++a) bias[a] = ba2 ? ba2[a] : 0.0f;
  bias[A] = bc2 ? bc2[0] : 0.0f;
  f32x4 pa = {0.0f, 0.0f, 0.0f, 0.0f}, pc = {0.0f, 0.0f, 0.0f, 0.0f};      // pre-activation biases of my 4 + 4 columns
  if (bac) {
    pa = *reinterpret_cast<const f32x4*>(bac + 4 * m.c4);
    pc = *reinterpret_cast<const f32x4*>(bac + C + 4 * m.c4);
  }
  const int64_t stride = (int64_t)gridDim.x * (kTB / 64) * m.rpw;
  const int64_t r0 = ((int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6)) * m.rpw + m.r_in;
  // every lane of a wave runs the same number of iterations (the butterfly needs all lanes)
  const int64_t rbase = ((int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6)) * m.rpw;
  for (int64_t rb = rbase, r = r0; rb < B; rb += stride, r += stride) {
    const bool live = r < B;
    const size_t o = (size_t)(live ? r : 0) * 2 * C + 4 * m.c4;
    f32x4 ha = {0.0f, 0.0f, 0.0f, 0.0f}, hc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
      ha = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Zac + o));
      hc = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Zac + o + C));
#pragma unroll
      for (int j = 0; j < 4; ++j) { ha[j] = train_tanhf(ha[j] + pa[j]); hc[j] = train_tanhf(hc[j] + pc[j]); }
      if (store_h) {      // store_h == 0: Zac keeps the pre-activations and heads_bwd recomputes the same tanh
        __builtin_nontemporal_store(ha, reinterpret_cast<f32x4*>(Zac + o));
        __builtin_nontemporal_store(hc, reinterpret_cast<f32x4*>(Zac + o + C));
      }
    }
    float p[A + 1];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s = fmaf(ha[j], wa[a][j], s);
      p[a] = s;
    }
    {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s = fmaf(hc[j], wc[j], s);
      p[A] = s;
    }
    for (int off = m.lpr >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int a = 0; a <= A; ++a) p[a] += __shfl_xor(p[a], off, 64);
    }
    if (live && m.c4 == 0) {
#pragma unroll
      for (int a = 0; a < A; ++a) logits[r * A + a] = p[a] + bias[a];
      value[r] = p[A] + bias[A];
    }
  }
}

// ------------------------------------------------------------------ B2' ------
__global__ __launch_bounds__(kTB) void tanh_bwd_colsum_kernel(float* __restrict__ dH, const float* __restrict__ H, int64_t B,
                                                              int C, float* __restrict__ partials) {
  extern __shared__ float sm[];
  const RowMap m(C, threadIdx.x & 63);
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int64_t stride = (int64_t)gridDim.x * (kTB / 64) * m.rpw;
  for (int64_t r = ((int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6)) * m.rpw + m.r_in; r < B; r += stride) {
    const size_t o = (size_t)r * C + 4 * m.c4;
    const f32x4 h = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(H + o));
    f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dH + o));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = g[j] * (1.0f - h[j] * h[j]);
      acc[j] += g[j];
    }
    __builtin_nontemporal_store(g, reinterpret_cast<f32x4*>(dH + o));
  }
  block_colsum<4>(acc, C, m, sm, partials);
}

// ------------------------------------------------------------------ B1 -------
// partial row layout: [db (C) | dW^T laid as d-major: d*C + col] -> (D+1) vectors of C columns
template <int D>
__global__ __launch_bounds__(kTB) void linear_smallk_bwd_kernel(const float* __restrict__ dH, const float* __restrict__ H,
                                                                const float* __restrict__ x, int64_t B, int C,
                                                                float* __restrict__ partials,
                                                                const float* __restrict__ W, const float* __restrict__ b) {
  extern __shared__ float sm[];
  const RowMap m(C, threadIdx.x & 63);
  // W != nullptr: H is not read — h = tanh(x W^T + b) is recomputed exactly as linear_tanh_smallk_kernel computed
  // it (same fmaf order, same fast_tanhf): 1 KB per row less traffic for 8 FMAs + a tanh per element
  float w[4][D], bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bias[j] = (W && b) ? b[4 * m.c4 + j] : 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) w[j][d] = W ? W[(size_t)(4 * m.c4 + j) * D + d] : 0.0f;
  }
  float acc[4 * (D + 1)];
#pragma unroll
  for (int i = 0; i < 4 * (D + 1); ++i) acc[i] = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * (kTB / 64) * m.rpw;
  for (int64_t r = ((int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6)) * m.rpw + m.r_in; r < B; r += stride) {
    const size_t o = (size_t)r * C + 4 * m.c4;
    const f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dH + o));
    float xv[D];
    load_row<D>(xv, x + r * D);
    f32x4 h;
    if (W) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) a = fmaf(xv[d], w[j][d], a);
        h[j] = train_tanhf(a + bias[j]);
      }
    } else if (H) {
      h = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(H + o));
    } else {
      h = f32x4{0.0f, 0.0f, 0.0f, 0.0f};          // dH is dZ already: g * (1 - 0) == g
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dz = g[j] * (1.0f - h[j] * h[j]);
      acc[j] += dz;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[4 * (d + 1) + j] = fmaf(dz, xv[d], acc[4 * (d + 1) + j]);
    }
  }
  block_colsum<4 * (D + 1)>(acc, C, m, sm, partials);
}

// dW[col][d] = sum over blocks of partials[b][(d+1)*C + col]; db[col] = ... [col]
__device__ __forceinline__ void smallk_finalize_body(const float* __restrict__ partials, int nblocks, int C, int D,
                                                     float* __restrict__ dW, float* __restrict__ db, int block) {
  const int e = block * kFinE + threadIdx.x % kFinE;
  const int width = (D + 1) * C;
  const double s = reduce_partials(partials, nblocks, width, e, e < width);
  if (e >= width || threadIdx.x >= kFinE) return;
  const int v = e / C, col = e - v * C;
  if (v == 0) db[col] = (float)s;
  else dW[(size_t)col * D + (v - 1)] = (float)s;
}
__global__ __launch_bounds__(kTB) void smallk_finalize_kernel(const float* __restrict__ partials, int nblocks, int C, int D,
                                                              float* __restrict__ dW, float* __restrict__ db) {
  smallk_finalize_body(partials, nblocks, C, D, dW, db, blockIdx.x);
}

// ------------------------------------------------------------------ B4 -------
// Hac [B, 2C] = [Ha | Hc] (tanh outputs of the actor / critic hidden layers).
// dZac [B, 2C] = [ (dlogits Wa2) * (1 - Ha^2) | (dv Wc2) * (1 - Hc^2) ]
// partial row: [dbac (2C) | dWa2 (A*C) | dWc2 (C) | dba2.. (C: only first A+1 used)]
template <int A>
__global__ __launch_bounds__(kTB) void heads_bwd_kernel(const float* Hac, const float* __restrict__ dlogits,
                                                        const float* __restrict__ dv, int64_t B, int C,
                                                        const float* __restrict__ Wa2, const float* __restrict__ Wc2,
                                                        float* dZac, float* __restrict__ partials,
                                                        int pre_activation, const float* __restrict__ bac) {
  extern __shared__ float sm[];
  const RowMap m(C, threadIdx.x & 63);
  const bool in_place = dZac == Hac;
  f32x4 pa = {0.0f, 0.0f, 0.0f, 0.0f}, pc = {0.0f, 0.0f, 0.0f, 0.0f};
  if (pre_activation && bac) {
    pa = *reinterpret_cast<const f32x4*>(bac + 4 * m.c4);
    pc = *reinterpret_cast<const f32x4*>(bac + C + 4 * m.c4);
  }
  float wa[A][4], wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wc[j] = Wc2[4 * m.c4 + j];
#pragma unroll
    for (int a = 0; a < A; ++a) wa[a][j] = Wa2[(size_t)a * C + 4 * m.c4 + j];
  }
  // accumulators: [0..3] dba (Ha cols), [4..7] dbc (Hc cols), [8 + 4a + j] dWa2[a], [8+4A + j] dWc2,
  // [12+4A + 0..3]: sums of dlogits / dv (A+1 values, lane-redundant; padded to a multiple of 4)
  constexpr int NACC = 8 + 4 * A + 4 + 4 * ((A + 1 + 3) / 4);
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * (kTB / 64) * m.rpw;
  for (int64_t r = ((int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6)) * m.rpw + m.r_in; r < B; r += stride) {
    const size_t o = (size_t)r * 2 * C + 4 * m.c4;
    f32x4 ha, hc;
    if (in_place) {       // dZac overwrites Hac: read-modify-write of the same lines, regular accesses (as tanh_inplace)
      ha = *reinterpret_cast<const f32x4*>(Hac + o);
      hc = *reinterpret_cast<const f32x4*>(Hac + o + C);
    } else {
      ha = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Hac + o));
      hc = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Hac + o + C));
    }
    if (pre_activation) {   // Hac holds actor.0 / critic.0 pre-activations: the forward's tanh, bit for bit
#pragma unroll
      for (int j = 0; j < 4; ++j) { ha[j] = train_tanhf(ha[j] + pa[j]); hc[j] = train_tanhf(hc[j] + pc[j]); }
    }
    float dl[A];
#pragma unroll
    for (int a = 0; a < A; ++a) dl[a] = dlogits[r * A + a];
    const float dvr = dv[r];
    f32x4 za, zc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int a = 0; a < A; ++a) {
        s = fmaf(dl[a], wa[a][j], s);
        acc[8 + 4 * a + j] = fmaf(dl[a], ha[j], acc[8 + 4 * a + j]);
      }
      za[j] = s * (1.0f - ha[j] * ha[j]);
      zc[j] = (dvr * wc[j]) * (1.0f - hc[j] * hc[j]);
      acc[j] += za[j];
      acc[4 + j] += zc[j];
      acc[8 + 4 * A + j] = fmaf(dvr, hc[j], acc[8 + 4 * A + j]);
    }
#pragma unroll
    for (int a = 0; a < A; ++a) acc[12 + 4 * A + a] += dl[a];
    acc[12 + 4 * A + A] += dvr;
    if (in_place) {
      *reinterpret_cast<f32x4*>(dZac + o) = za;
      *reinterpret_cast<f32x4*>(dZac + o + C) = zc;
    } else {
      __builtin_nontemporal_store(za, reinterpret_cast<f32x4*>(dZac + o));
      __builtin_nontemporal_store(zc, reinterpret_cast<f32x4*>(dZac + o + C));
    }
  }
  // only the c4 == 0 lane of a row group contributes the dlogits / dv sums (they are lane-redundant)
  if (m.c4 != 0) {
#pragma unroll
    for (int i = 12 + 4 * A; i < NACC; ++i) acc[i] = 0.0f;
  }
  block_colsum<NACC>(acc, C, m, sm, partials);
}

// partial row (vectors of C): v0 dba, v1 dbc, v2..v(1+A) dWa2[a], v(2+A) dWc2, v(3+A).. dlogit sums at col 0..3
template <int A>
__device__ __forceinline__ void heads_finalize_body(const float* __restrict__ partials, int nblocks, int C,
                                                    float* __restrict__ dbac, float* __restrict__ dWa2,
                                                    float* __restrict__ dWc2, float* __restrict__ dba2,
                                                    float* __restrict__ dbc2, int block) {
  constexpr int NV = (8 + 4 * A + 4 + 4 * ((A + 1 + 3) / 4)) / 4;
  const int width = NV * C;
  const int e = block * kFinE + threadIdx.x % kFinE;
  const int v = e / C, col = e - v * C;
  const bool want = e < width && !(v >= 3 + A && col >= 4);
  const double s = reduce_partials(partials, nblocks, width, e, want);
  if (!want || threadIdx.x >= kFinE) return;
  const float f = (float)s;
  if (v == 0) dbac[col] = f;
  else if (v == 1) dbac[C + col] = f;
  else if (v < 2 + A) dWa2[(size_t)(v - 2) * C + col] = f;
  else if (v == 2 + A) dWc2[col] = f;
  else {
    const int k = 4 * (v - 3 - A) + col;       // index into (dlogit sums..., dv sum)
    if (k < A) dba2[k] = f;
    else if (k == A) dbc2[0] = f;
  }
}
template <int A>
__global__ __launch_bounds__(kTB) void heads_finalize_kernel(const float* __restrict__ partials, int nblocks, int C,
                                                             float* __restrict__ dbac, float* __restrict__ dWa2,
                                                             float* __restrict__ dWc2, float* __restrict__ dba2,
                                                             float* __restrict__ dbc2) {
  heads_finalize_body<A>(partials, nblocks, C, dbac, dWa2, dWc2, dba2, dbc2, blockIdx.x);
}

// ------------------------------------------------------------------ F3+L1+B4 -
// heads_fwd_tanh + ppo_loss + heads_bwd as ONE pass over Zac [B, 512] (pre-activations of actor.0 | critic.0,
// C = 256: a row is one wavefront x float4 + float4).  A wave takes kRB rows at a time: tanh, the A + 1 head dot
// products reduced with the forward pass's butterfly, lane k keeps row k's logits and value; then ONE evaluation of
// the clipped-surrogate loss for the kRB rows (ppo_loss_row, one row per lane: the bits of ppo_loss_kernel — done
// per row it would be 250 instructions replicated over the row's 64 lanes, more than the rest of the pass); then row
// k's dlogits / dv are read back from lane k (v_readlane) and turned into dZac — written in place over the
// pre-activations — and into heads_bwd's register accumulators.  One read and one write of 2 KB per row instead of
// two reads and a write plus the logits / dlogits round trip.  partial row: as heads_bwd_kernel;
// met_parts[block][5] (f64).
constexpr int kRB = 8;
__device__ __forceinline__ float lane_bcast(float v, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
template <int A>
__global__ __launch_bounds__(kTB) void heads_loss_kernel(float* Zac, int64_t B, const float* __restrict__ bac,
                                                         const float* __restrict__ Wa2, const float* __restrict__ ba2,
                                                         const float* __restrict__ Wc2, const float* __restrict__ bc2,
                                                         const int32_t* __restrict__ act, const float* __restrict__ logp_old,
                                                         const float* __restrict__ adv, const float* __restrict__ ret,
                                                         const double* __restrict__ adv_moments, gymrl_ppo_cfg cfg,
                                                         float* __restrict__ partials, double* __restrict__ met_parts) {
  constexpr int C = 256;
  extern __shared__ float sm[];
  __shared__ double s_norm[2];
  __shared__ double s_met[5][kTB / 64];
  if (threadIdx.x == 0) {
    double mean = 0.0, sd = 1.0;
    if (adv_moments) {      // whole-rollout advantage normalisation (ppo_lunarlander.py:236), as ppo_loss_kernel
      const double cnt = adv_moments[0];
      mean = adv_moments[1] / cnt;
      double var = adv_moments[2] / cnt - mean * mean;
      var = var > 0.0 ? var : 0.0;
      sd = sqrt(var) + 1e-8;
    }
    s_norm[0] = mean; s_norm[1] = sd;
  }
  __syncthreads();
  const RowMap m(C, threadIdx.x & 63);               // lpr = 64: c4 = lane, one row per wave-load
  const int lane = threadIdx.x & 63;
  float wa[A][4], wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wc[j] = Wc2[4 * m.c4 + j];
#pragma unroll
    for (int a = 0; a < A; ++a) wa[a][j] = Wa2[(size_t)a * C + 4 * m.c4 + j];
  }
  float hb[A + 1];
#pragma unroll
  for (int a = 0; a < A; ++a) hb[a] = ba2 ? ba2[a] : 0.0f;
  hb[A] = bc2 ? bc2[0] : 0.0f;
  f32x4 pa = {0.0f, 0.0f, 0.0f, 0.0f}, pc = {0.0f, 0.0f, 0.0f, 0.0f};
  if (bac) {
    pa = *reinterpret_cast<const f32x4*>(bac + 4 * m.c4);
    pc = *reinterpret_cast<const f32x4*>(bac + C + 4 * m.c4);
  }
  constexpr int NACC = 8 + 4 * A + 4 + 4 * ((A + 1 + 3) / 4);
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
  double met[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const float invB = 1.0f / (float)B;
  const int64_t nbatch = (B + kRB - 1) / kRB;
  for (int64_t bt = (int64_t)blockIdx.x * (kTB / 64) + (threadIdx.x >> 6); bt < nbatch; bt += (int64_t)gridDim.x * (kTB / 64)) {
    const int64_t r0 = bt * kRB;
    f32x4 ha[kRB], hc[kRB];
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
      const int64_t rr = r0 + k < B ? r0 + k : B - 1;
      const size_t o = (size_t)rr * 2 * C + 4 * m.c4;
      ha[k] = *reinterpret_cast<const f32x4*>(Zac + o);
      hc[k] = *reinterpret_cast<const f32x4*>(Zac + o + C);
    }
    // the loss inputs of row r0 + lane (lanes >= kRB and rows past the end: a valid row, results unused)
    const int64_t rl = (lane < kRB && r0 + lane < B) ? r0 + lane : r0;
    const int a_r = act[rl];
    const float lpo = logp_old[rl], rt = ret[rl];
    float ad = adv[rl];
    float z[A], v = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) z[a] = 0.0f;
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ha[k][j] = train_tanhf(ha[k][j] + pa[j]); hc[k][j] = train_tanhf(hc[k][j] + pc[j]); }
      float p[A + 1];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(ha[k][j], wa[a][j], s);
        p[a] = s;
      }
      {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(hc[k][j], wc[j], s);
        p[A] = s;
      }
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a <= A; ++a) p[a] += __shfl_xor(p[a], off, 64);
      }
      if (lane == k) {
#pragma unroll
        for (int a = 0; a < A; ++a) z[a] = p[a] + hb[a];
        v = p[A] + hb[A];
      }
    }
    float dl[A], dvr, m_obj, m_val, m_ent, m_clip, m_kl;
    if (adv_moments) ad = (float)(((double)ad - s_norm[0]) / s_norm[1]);
    ppo_loss_row<A>(z, v, a_r, lpo, ad, rt, invB, cfg, dl, dvr, m_obj, m_val, m_ent, m_clip, m_kl);
    if (lane < kRB && r0 + lane < B) {
      met[0] += -(double)m_obj; met[1] += (double)m_val; met[2] += (double)m_ent;
      met[3] += (double)m_clip; met[4] += (double)m_kl;
    }
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
      if (r0 + k < B) {                      // wave-uniform
        float dk[A];
#pragma unroll
        for (int a = 0; a < A; ++a) dk[a] = lane_bcast(dl[a], k);
        const float dvk = lane_bcast(dvr, k);
        f32x4 za, zc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = 0.0f;
#pragma unroll
          for (int a = 0; a < A; ++a) {
            s = fmaf(dk[a], wa[a][j], s);
            acc[8 + 4 * a + j] = fmaf(dk[a], ha[k][j], acc[8 + 4 * a + j]);
          }
          za[j] = s * (1.0f - ha[k][j] * ha[k][j]);
          zc[j] = (dvk * wc[j]) * (1.0f - hc[k][j] * hc[k][j]);
          acc[j] += za[j];
          acc[4 + j] += zc[j];
          acc[8 + 4 * A + j] = fmaf(dvk, hc[k][j], acc[8 + 4 * A + j]);
        }
#pragma unroll
        for (int a = 0; a < A; ++a) acc[12 + 4 * A + a] += dk[a];
        acc[12 + 4 * A + A] += dvk;
        const size_t o = (size_t)(r0 + k) * 2 * C + 4 * m.c4;
        *reinterpret_cast<f32x4*>(Zac + o) = za;
        *reinterpret_cast<f32x4*>(Zac + o + C) = zc;
      }
    }
  }
  if (m.c4 != 0) {
#pragma unroll
    for (int i = 12 + 4 * A; i < NACC; ++i) acc[i] = 0.0f;
  }
  // metrics: wave sums -> LDS -> one f64 row per workgroup
  {
    const int wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const double s = wave_sum(met[k]);
      if (lane == 0) s_met[k][wid] = s;
    }
  }
  block_colsum<NACC>(acc, C, m, sm, partials);          // (its barrier also publishes s_met)
  if (threadIdx.x < 5) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kTB / 64; ++w) s += s_met[threadIdx.x][w];
    met_parts[(size_t)blockIdx.x * 5 + threadIdx.x] = s;
  }
}

inline bool pow2_cols(int C) { return C >= 16 && C <= 256 && (C & (C - 1)) == 0; }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline size_t sm_bytes(int C, int nv4) { return sizeof(float) * (size_t)(kTB / 64) * (64 / (C / 4)) * nv4 * C; }


// ------------------------------------------------------------------ the minibatch's reductions, one launch ---
// A minibatch's backward leaves five batch reductions half done — the slice partials of the two 256-deep weight gradients
// (gemm.hip), the block partials of the heads' and of the first layer's gradients (above) — and their second halves were
// five launches of 5-17 us behind one another (3 % of the update).  They do not depend on each other: here they are block
// ranges of ONE grid, each block running the body its own kernel runs (finalize_device.hpp, *_finalize_body): same loads,
// same float64 sums in the same order, bit-identical outputs.
struct UpdFinArgs {
  const float* parts_ac; int slices_ac; float* dWac;                    // [2C x C] from 2C/256 tiles x slices_ac partials
  const float* parts_2; const float* cs_2; int slices_2; float* dW2; float* db2;
  const float* parts_h; int nb_h; float* dbac; float* dWa2; float* dWc2; float* dba2; float* dbc2;
  const float* parts_1; int nb_1; int D; float* dW1; float* db1;
  int C, b_ac, b_2, b_cs, b_h;                                           // first block of each job after the first
};
template <int A>
__global__ __launch_bounds__(kTB) void update_finalize_kernel(const UpdFinArgs a) {
  __shared__ double sm[3][64][4];
  const int b = blockIdx.x;
  if (b < a.b_ac) fin::tn_reduce_body(a.parts_ac, a.slices_ac, a.C, a.dWac, b, sm);
  else if (b < a.b_2) fin::tn_reduce_body(a.parts_2, a.slices_2, a.C, a.dW2, b - a.b_ac, sm);
  else if (b < a.b_cs) fin::tn_colsum_body(a.cs_2, a.slices_2, a.db2, b - a.b_2, reinterpret_cast<double (*)[64]>(sm));
  else if (b < a.b_h) heads_finalize_body<A>(a.parts_h, a.nb_h, a.C, a.dbac, a.dWa2, a.dWc2, a.dba2, a.dbc2, b - a.b_cs);
  else smallk_finalize_body(a.parts_1, a.nb_1, a.C, a.D, a.dW1, a.db1, b - a.b_h);
}

}  // namespace

extern "C" {

size_t gymrl_mlp_train_workspace_bytes(int C, int D, int A) {
  if (C <= 0 || D < 0 || A < 0) return 0;
  const size_t w1 = (size_t)(D + 1) * C, w2 = (size_t)(3 + A + (A + 1 + 3) / 4) * C;
  return sizeof(float) * kMaxBlocks * (w1 > w2 ? w1 : w2) + 256;
}

int gymrl_linear_tanh_smallk(const float* x, const float* W, const float* b, int64_t B, int D, int C, float* out,
                             void* stream) {
  if (!x || !W || !out || B < 0 || !pow2_cols(C) || !al16(out) || !al16(x)) return -22;
  if (B == 0) return 0;
  const dim3 grid(grid_for(B, C) * 1), block(kTB);
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 2: hipLaunchKernelGGL(linear_tanh_smallk_kernel<2>, grid, block, 0, s, x, W, b, B, C, out); break;
    case 3: hipLaunchKernelGGL(linear_tanh_smallk_kernel<3>, grid, block, 0, s, x, W, b, B, C, out); break;
    case 4: hipLaunchKernelGGL(linear_tanh_smallk_kernel<4>, grid, block, 0, s, x, W, b, B, C, out); break;
    case 8: hipLaunchKernelGGL(linear_tanh_smallk_kernel<8>, grid, block, 0, s, x, W, b, B, C, out); break;
    default: return -22;
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_smallk(const float* x, const float* W, const float* b, int64_t B, int D, int C, float* out, void* stream) {
  if (!x || !W || !out || B < 0 || !pow2_cols(C) || !al16(out) || !al16(x)) return -22;
  if (B == 0) return 0;
  const dim3 grid(grid_for(B, C) * 1), block(kTB);
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 2: hipLaunchKernelGGL((linear_tanh_smallk_kernel<2, false>), grid, block, 0, s, x, W, b, B, C, out); break;
    case 3: hipLaunchKernelGGL((linear_tanh_smallk_kernel<3, false>), grid, block, 0, s, x, W, b, B, C, out); break;
    case 4: hipLaunchKernelGGL((linear_tanh_smallk_kernel<4, false>), grid, block, 0, s, x, W, b, B, C, out); break;
    case 8: hipLaunchKernelGGL((linear_tanh_smallk_kernel<8, false>), grid, block, 0, s, x, W, b, B, C, out); break;
    default: return -22;
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_tanh_inplace(float* z, int64_t n, const float* bias, int C, void* stream) {
  if (!z || n < 0 || (n & 3) || !al16(z)) return -22;
  if (bias && (C < 4 || (C & (C - 1)) || (n % C) || !al16(bias))) return -22;
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  int64_t nb = (n4 + kTB - 1) / kTB;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(tanh_inplace_kernel, dim3((unsigned)nb), dim3(kTB), 0, (hipStream_t)stream, z, n4, bias, bias ? C / 4 : 1);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_tanh_bwd_colsum(float* dH, const float* H, int64_t B, int C, float* colsum_out, void* workspace,
                          void* stream) {
  if (!dH || !H || !colsum_out || !workspace || B < 0 || !pow2_cols(C) || !al16(dH) || !al16(H)) return -22;
  hipStream_t s = (hipStream_t)stream;
  const int nb = grid_for(B, C);
  float* parts = (float*)workspace;
  hipLaunchKernelGGL(tanh_bwd_colsum_kernel, dim3(nb), dim3(kTB), sm_bytes(C, 1), s, dH, H, B, C, parts);
  hipLaunchKernelGGL(smallk_finalize_kernel, dim3((C + kFinE - 1) / kFinE), dim3(kTB), 0, s, parts, nb, C, 0, (float*)nullptr,
                     colsum_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_smallk_bwd(const float* dH, const float* H, const float* x, int64_t B, int D, int C, float* dW,
                            float* db, const float* W, const float* b, void* workspace, void* stream) {
  // dW == NULL && db == NULL: the block partials only (they stay in `workspace` for gymrl_update_finalize)
  if (!dH || !x || (!dW != !db) || !workspace || B < 0 || !pow2_cols(C) || !al16(dH) || (H && !al16(H)) ||
      !al16(x))
    return -22;
  hipStream_t s = (hipStream_t)stream;
  const int nb = grid_for(B, C);
  float* parts = (float*)workspace;
  const dim3 grid(nb), block(kTB);
  switch (D) {
    case 2: hipLaunchKernelGGL(linear_smallk_bwd_kernel<2>, grid, block, sm_bytes(C, 3), s, dH, H, x, B, C, parts, W, b); break;
    case 3: hipLaunchKernelGGL(linear_smallk_bwd_kernel<3>, grid, block, sm_bytes(C, 4), s, dH, H, x, B, C, parts, W, b); break;
    case 4: hipLaunchKernelGGL(linear_smallk_bwd_kernel<4>, grid, block, sm_bytes(C, 5), s, dH, H, x, B, C, parts, W, b); break;
    case 8: hipLaunchKernelGGL(linear_smallk_bwd_kernel<8>, grid, block, sm_bytes(C, 9), s, dH, H, x, B, C, parts, W, b); break;
    default: return -22;
  }
  if (dW) hipLaunchKernelGGL(smallk_finalize_kernel, dim3(((D + 1) * C + kFinE - 1) / kFinE), dim3(kTB), 0, s, parts, nb, C, D, dW, db);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_heads_fwd_tanh(float* Zac, int64_t B, int C, int A, const float* bac, const float* Wa2, const float* ba2,
                         const float* Wc2, const float* bc2, float* logits, float* value, int store_h, void* stream) {
  if (!Zac || !Wa2 || !Wc2 || !logits || !value || B < 0 || !pow2_cols(C) || !al16(Zac) || (bac && !al16(bac))) return -22;
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // read-only stream with a per-row butterfly: latency-bound per wave, so more resident waves than the kernels
  // that write block partials (no workspace bounds the grid here)
  const int64_t rows_per_block = (kTB / 64) * (64 / (C / 4));
  const int64_t want = (B + rows_per_block - 1) / rows_per_block;
  const dim3 grid((unsigned)(want < 8 * kMaxBlocks ? (want < 1 ? 1 : want) : 8 * kMaxBlocks)), block(kTB);
  if (A == 4) hipLaunchKernelGGL(heads_fwd_tanh_kernel<4>, grid, block, 0, s, Zac, B, C, bac, Wa2, ba2, Wc2, bc2, logits, value, store_h);
  else if (A == 2) hipLaunchKernelGGL(heads_fwd_tanh_kernel<2>, grid, block, 0, s, Zac, B, C, bac, Wa2, ba2, Wc2, bc2, logits, value, store_h);
  else return -22;
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_heads_bwd(const float* Hac, const float* dlogits, const float* dv, int64_t B, int C, int A,
                    const float* Wa2, const float* Wc2, float* dZac, float* dbac, float* dWa2, float* dba2,
                    float* dWc2, float* dbc2, int pre_activation, const float* bac, void* workspace, void* stream) {
  if (!Hac || !dlogits || !dv || !Wa2 || !Wc2 || !dZac || !dbac || !dWa2 || !dba2 || !dWc2 || !dbc2 || !workspace ||
      B < 0 || !pow2_cols(C) || !al16(Hac) || !al16(dZac))
    return -22;
  hipStream_t s = (hipStream_t)stream;
  const int nb = grid_for(B, C);
  float* parts = (float*)workspace;
  const dim3 grid(nb), block(kTB);
  if (A == 4) {
    constexpr int NV = (8 + 16 + 4 + 8) / 4;
    hipLaunchKernelGGL(heads_bwd_kernel<4>, grid, block, sm_bytes(C, NV), s, Hac, dlogits, dv, B, C, Wa2, Wc2, dZac, parts, pre_activation, bac);
    hipLaunchKernelGGL(heads_finalize_kernel<4>, dim3((NV * C + kFinE - 1) / kFinE), block, 0, s, parts, nb, C, dbac, dWa2,
                       dWc2, dba2, dbc2);
  } else if (A == 2) {
    constexpr int NV = (8 + 8 + 4 + 4) / 4;
    hipLaunchKernelGGL(heads_bwd_kernel<2>, grid, block, sm_bytes(C, NV), s, Hac, dlogits, dv, B, C, Wa2, Wc2, dZac, parts, pre_activation, bac);
    hipLaunchKernelGGL(heads_finalize_kernel<2>, dim3((NV * C + kFinE - 1) / kFinE), block, 0, s, parts, nb, C, dbac, dWa2,
                       dWc2, dba2, dbc2);
  } else {
    return -22;
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

static int heads_loss_grid(int64_t B) {
  const int64_t nb = ((B + kRB - 1) / kRB + (kTB / 64) - 1) / (kTB / 64);
  return (int)(nb < kMaxBlocks ? (nb < 1 ? 1 : nb) : kMaxBlocks);
}

int gymrl_heads_loss_blocks(int64_t B, int C) {
  if (B <= 0 || C != 256) return 0;
  return heads_loss_grid(B);
}

int gymrl_heads_loss_fwd_bwd(float* Zac, int64_t B, int C, int A, const float* bac, const float* Wa2, const float* ba2,
                             const float* Wc2, const float* bc2, const int32_t* act, const float* logp_old,
                             const float* adv, const float* ret, const double* adv_moments, const gymrl_ppo_cfg* cfg,
                             float* dbac, float* dWa2, float* dba2, float* dWc2, float* dbc2, double* metric_parts,
                             void* workspace, void* stream) {
  // all five gradient outputs NULL: the block partials only (they stay in `workspace` for gymrl_update_finalize)
  const bool parts_only = !dbac && !dWa2 && !dba2 && !dWc2 && !dbc2;
  if (!Zac || !Wa2 || !Wc2 || !act || !logp_old || !adv || !ret || !cfg ||
      (!parts_only && (!dbac || !dWa2 || !dba2 || !dWc2 || !dbc2)) ||
      !metric_parts || !workspace || B <= 0 || C != 256 || !al16(Zac) || (bac && !al16(bac)))
    return -22;
  hipStream_t s = (hipStream_t)stream;
  const int nb = heads_loss_grid(B);
  float* parts = (float*)workspace;
  const dim3 grid(nb), block(kTB);
  if (A == 4) {
    constexpr int NV = (8 + 16 + 4 + 8) / 4;
    hipLaunchKernelGGL(heads_loss_kernel<4>, grid, block, sm_bytes(C, NV), s, Zac, B, bac, Wa2, ba2, Wc2, bc2, act,
                       logp_old, adv, ret, adv_moments, *cfg, parts, metric_parts);
    if (!parts_only)
      hipLaunchKernelGGL(heads_finalize_kernel<4>, dim3((NV * C + kFinE - 1) / kFinE), block, 0, s, parts, nb, C, dbac, dWa2,
                         dWc2, dba2, dbc2);
  } else if (A == 2) {
    constexpr int NV = (8 + 8 + 4 + 4) / 4;
    hipLaunchKernelGGL(heads_loss_kernel<2>, grid, block, sm_bytes(C, NV), s, Zac, B, bac, Wa2, ba2, Wc2, bc2, act,
                       logp_old, adv, ret, adv_moments, *cfg, parts, metric_parts);
    if (!parts_only)
      hipLaunchKernelGGL(heads_finalize_kernel<2>, dim3((NV * C + kFinE - 1) / kFinE), block, 0, s, parts, nb, C, dbac, dWa2,
                         dWc2, dba2, dbc2);
  } else {
    return -22;
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_update_finalize(int64_t B, int C, int A, int D, const void* ws_dw_ac, float* dWac, const void* ws_dw_2, float* dW2,
                          float* db2, const void* ws_heads, float* dbac, float* dWa2, float* dba2, float* dWc2, float* dbc2,
                          const void* ws_smallk, float* dW1, float* db1, void* stream) {
  if (B <= 0 || C != 256 || (A != 2 && A != 4) || (D != 2 && D != 3 && D != 4 && D != 8) || !ws_dw_ac || !dWac || !ws_dw_2 ||
      !dW2 || !db2 || !ws_heads || !dbac || !dWa2 || !dba2 || !dWc2 || !dbc2 || !ws_smallk || !dW1 || !db1 || !al16(dWac) ||
      !al16(dW2) || !al16(ws_dw_ac) || !al16(ws_dw_2))
    return -22;
  UpdFinArgs a{};
  int64_t rps;
  fin::tn_geometry(B, 2 * C, &a.slices_ac, &rps);
  fin::tn_geometry(B, C, &a.slices_2, &rps);
  a.parts_ac = (const float*)((const char*)ws_dw_ac + fin::kColsumBytes); a.dWac = dWac;
  a.cs_2 = (const float*)ws_dw_2; a.parts_2 = (const float*)((const char*)ws_dw_2 + fin::kColsumBytes); a.dW2 = dW2; a.db2 = db2;
  a.parts_h = (const float*)ws_heads; a.nb_h = heads_loss_grid(B);
  a.dbac = dbac; a.dWa2 = dWa2; a.dWc2 = dWc2; a.dba2 = dba2; a.dbc2 = dbc2;
  a.parts_1 = (const float*)ws_smallk; a.nb_1 = grid_for(B, C); a.D = D; a.dW1 = dW1; a.db1 = db1;
  a.C = C;
  const int NV = (8 + 4 * A + 4 + 4 * ((A + 1 + 3) / 4)) / 4;
  a.b_ac = (2 * C / 256) * 256;
  a.b_2 = a.b_ac + (C / 256) * 256;
  a.b_cs = a.b_2 + (C / 256) * 4;
  a.b_h = a.b_cs + (NV * C + kFinE - 1) / kFinE;
  const int total = a.b_h + ((D + 1) * C + kFinE - 1) / kFinE;
  if (A == 4) hipLaunchKernelGGL(update_finalize_kernel<4>, dim3(total), dim3(kTB), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(update_finalize_kernel<2>, dim3(total), dim3(kTB), 0, (hipStream_t)stream, a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
