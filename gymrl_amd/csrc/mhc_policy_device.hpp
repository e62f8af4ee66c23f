// mhc_policy_device.hpp — PPO-full's ActorCritic.forward (:377-407) for 16 rows on one 256-thread workgroup, shared by
// mhc_policy_kernel (csrc/mhc.hip: the step-by-step rollout forward) and the persistent rollout (csrc/rollout_lunar.hip), plus the
// transcendental helpers every mHC kernel uses.
#pragma once
#include "train_device.hpp"
#include "../../include/gymrl.h"

namespace gymrl {
namespace mhc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// exp and 1/x on the hardware units (v_exp_f32, v_rcp_f32: 1 ulp each).  The gate arithmetic and the SiLUs are what these kernels
// issue most — a correctly rounded division is ~12 instructions, libm's expf ~15 — and every result is held to 1e-5 of the
// float64 modules, not to torch's bits.  exp_: x log2(e) in two pieces, so that the product's rounding (up to |x| 2^-24 relative
// in the result) is folded back in.
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float exp_(float x) {
  const float t = x * 1.44269504088896341f;
  const float lo = fmaf(x, 1.44269504088896341f, -t) + x * 1.92596299112661746e-8f;
  return __builtin_amdgcn_exp2f(t) * (1.0f + lo * 0.693147180559945309f);
}
__device__ __forceinline__ float sigmoidf_(float x) { return rcp_(1.0f + exp_(-x)); }
__device__ __forceinline__ float silu_(float z) { return z * sigmoidf_(z); }
__device__ __forceinline__ float silu_grad_(float z) { const float s = sigmoidf_(z); return s * (1.0f + z * (1.0f - s)); }

// sum over the 16 lanes of a DPP row, every lane ending with the same bits: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror — four v_add_f32_dpp, no LDS traffic (a 64-lane __shfl_xor tree is six ds_bpermute round trips)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  return v;
}

// ---- the whole rollout forward of PPO-full's network in ONE launch ------------------------------------------------------
// ActorCritic.forward (:377-407) for n = 2 branches of D = 128 and 256-wide heads: input projection, every hyper-connection
// sub-block, final_norm(h.sum(1)), both heads (Linear -> SiLU -> RMSNorm -> Linear).  Nothing couples two rows of the batch,
// so a workgroup of four waves carries 16 rows through the network: the branch stack stays in registers from the first
// layer to the last (lane (row, sub) holds columns 64 q + 4 sub .. + 3 of its row, as in mhc_gates2_kernel), only the
// 16 x 128 operand of each Linear and its output cross LDS, and the waves split the Linear's column tiles
// (v_mfma_f32_16x16x4_f32, weights from L2).  As 19 launches the forward is 177 us per 4096-row vector step (each launch
// 6-14 us of latency: 256 waves on 1024 SIMDs); the per-row work is ~30 us.
constexpr int kPolMaxSub = 8, kPolMaxOut = 8, kPolPad = 132;
// Probe build (make prof): 100 MHz ticks workgroup 0's first lane spends in the phases of policy_tile, summed over the calls
// ([15] counts them): 0 input projection, 1 gates + read (all sub-blocks), 2 Linear + SiLU, 3 combine, 4 final norm, 5 heads, 6 tail
#ifdef GYMRL_LUNAR_PROF
static __device__ unsigned long long g_pol_prof[16];
#define POL_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); g_pol_prof[i] += now_ - pol_t_; pol_t_ = now_; } } while (0)
#else
#define POL_MARK(i) do {} while (0)
#endif
struct PolicyArgs {
  const float* in_w; const float* in_b;
  const float* norm_w[kPolMaxSub]; const float* gw[kPolMaxSub]; const float* alpha[kPolMaxSub]; const float* beta[kPolMaxSub];
  const float* lw[kPolMaxSub]; const float* lb[kPolMaxSub];
  const float* fn_w;
  const float* h1_w[2]; const float* h1_b[2]; const float* hn_w[2]; const float* h2_w[2]; const float* h2_b[2];
  float fn_eps, hn_eps[2];
  int obs_dim, n_sub, n_act, sk_it;
  const float* img;                                        // gymrl_mhc_policy_pack's image of the wide operands, or nullptr: read in place
};

// The image: per sub-block its Linear [128 x 128] as MFMA B-operand tiles — (tile T, k-step j, lane) -> W[16 T + lane % 16][16 j + 4 (lane / 16) .. + 3],
// one contiguous KiB per wave-wide load — then its gate weights in the order the row view reads them ((q, e, half, sub) ->
// w[64 q + 4 sub + e][4 half .. + 3]); after the sub-blocks the two heads' first Linears [256 x 128] as 16 tiles each.  Read from
// nn.Linear's rows a wave-wide 16-byte load touches sixteen cache lines (a 64-byte piece of sixteen rows) and the compute unit's
// memory pipe moves the 0.5 MB a step needs at 21 GB/s; from the image at 50+ (tools/probe_mhc_policy.py).  Same values.
constexpr int kPolLwFloats = 128 * 128, kPolGwFloats = 256 * 8, kPolSubFloats = kPolLwFloats + kPolGwFloats, kPolHeadFloats = 256 * 128;
__host__ __device__ inline size_t policy_image_floats(int n_sub) { return (size_t)n_sub * kPolSubFloats + 2 * (size_t)kPolHeadFloats; }

struct PolicyLds {
  float rbuf[16][kPolPad];                                 // a Linear's input rows (the MFMA A operand)
  float obuf[16][kPolPad];                                 // its activated output rows
  float psq[4][16];                                        // heads: per wave (head, half) the rows' sums of squares
  float pdot[4][16][kPolMaxOut];                           //        and partial output dot products
};

// 16 rows through the network on a 256-thread workgroup.  obs_row: THIS lane's row (row 4 wave + lane / 16 of the tile; global
// memory or LDS); logits_out [16][ld_logits], value_out [16 * ld_value] (global or LDS), written for rows < n_valid by threads
// 0..31 — a barrier is the caller's business if other threads read them.
__device__ __forceinline__ void policy_tile(const PolicyArgs& a, PolicyLds& L, const float* obs_row, float* logits_out, int ld_logits,
                                            float* value_out, int ld_value, int n_valid) {
  constexpr int D = 128, NC = 256, G = 8, H = 256;
  float (&rbuf)[16][kPolPad] = L.rbuf;
  float (&obuf)[16][kPolPad] = L.obuf;
  float (&psq)[4][16] = L.psq;
  float (&pdot)[4][16][kPolMaxOut] = L.pdot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, grp = lane >> 4;              // row view: lane = (row 4 wave + grp, 16-column slot sub)
  const int r = sub, qq = grp;                             // MFMA view: lane = (row / column r, k-quarter qq)
  const int lrow = 4 * wave + grp;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#ifdef GYMRL_LUNAR_PROF
  unsigned long long pol_t_ = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) g_pol_prof[15] += 1;
#endif

  // input projection (:178-181): z = obs W^T + b, both branches start as z
  f32x4 x[4];
  if (a.obs_dim == 8 && ((reinterpret_cast<uintptr_t>(a.in_w) | reinterpret_cast<uintptr_t>(a.in_b) | reinterpret_cast<uintptr_t>(obs_row)) & 15) == 0) {
    // LunarLander's width: the lane's eight weight rows as sixteen 16-byte loads and its two bias quads, all requested before the
    // first product (the general form below issues one dword load per product behind a test of obs_dim: 8.5 us of a 58 us
    // forward, tools/probe_mhc_policy.py; 2.6 us this way).  Same products, same order: z = b, then + ob[k] w[k] for k = 0 .. 7.
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(obs_row), o1 = *reinterpret_cast<const f32x4*>(obs_row + 4);
    f32x4 w[2][4][2], bq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bq[q] = *reinterpret_cast<const f32x4*>(a.in_b + 64 * q + 4 * sub);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wr = a.in_w + (size_t)(64 * q + 4 * sub + e) * 8;
        w[q][e][0] = *reinterpret_cast<const f32x4*>(wr); w[q][e][1] = *reinterpret_cast<const f32x4*>(wr + 4);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = bq[q][e];
#pragma unroll
        for (int k = 0; k < 4; ++k) z += o0[k] * w[q][e][0][k];
#pragma unroll
        for (int k = 0; k < 4; ++k) z += o1[k] * w[q][e][1][k];
        x[q][e] = z;
      }
    x[2] = x[0]; x[3] = x[1];
  } else {
    float ob[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ob[k] = k < a.obs_dim ? obs_row[k] : 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 64 * q + 4 * sub + e;
        float z = a.in_b[c];
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < a.obs_dim) z += ob[k] * a.in_w[c * a.obs_dim + k];
        x[q][e] = z;
      }
    x[2] = x[0]; x[3] = x[1];
  }
  POL_MARK(0);

  for (int s = 0; s < a.n_sub; ++s) {
    // gates (:125-147): every lane of a row's 16 ends with the row's nine sums and evaluates the gates itself
    float Hs[G + 1];
#pragma unroll
    for (int k = 0; k <= G; ++k) Hs[k] = 0.0f;
    const float* __restrict__ nwp = a.norm_w[s];
    const float* __restrict__ gwp = a.gw[s];
    // The lane's gate weights in two rounds of 18 sixteen-byte loads, each round requested before its first product: with the
    // loads inside the (q, e) loop every one of the 16 iterations waited out its own L2 round trip — 6 of the sub-block's 7 us
    // (tools/probe_mhc_policy.py).  Same products in the same order.
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      f32x4 nw[2], lo[2][4], hi[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = 64 * (2 * hq + u) + 4 * sub;
        nw[u] = *reinterpret_cast<const f32x4*>(nwp + c);
        if (a.img) {
          const f32x4* gi = reinterpret_cast<const f32x4*>(a.img + (size_t)s * kPolSubFloats + kPolLwFloats) + (size_t)(2 * hq + u) * 128 + sub;
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo[u][e] = gi[e * 32]; hi[u][e] = gi[e * 32 + 16]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lo[u][e] = *reinterpret_cast<const f32x4*>(gwp + (size_t)(c + e) * G);
            hi[u][e] = *reinterpret_cast<const f32x4*>(gwp + (size_t)(c + e) * G + 4);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = 2 * hq + u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xv = x[q][e], t = nw[u][e] * xv;
          Hs[G] += xv * xv;
#pragma unroll
          for (int k = 0; k < 4; ++k) { Hs[k] += t * lo[u][e][k]; Hs[4 + k] += t * hi[u][e][k]; }
        }
      }
    }
    // this wave's two weight tiles of the sub-block's Linear: requested behind the gates' operands and ahead of the reductions,
    // transcendentals and Sinkhorn sweeps whose latency hides their arrival
    POL_MARK(7);
    f32x4 wv[2][8];
    if (a.img) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4* wi = reinterpret_cast<const f32x4*>(a.img + (size_t)s * kPolSubFloats) + (size_t)(2 * wave + t) * 512 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[t][j] = wi[64 * j];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* wrow = a.lw[s] + (size_t)(16 * (2 * wave + t) + r) * D + 4 * qq;
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[t][j] = *reinterpret_cast<const f32x4*>(wrow + 16 * j);
      }
    }
#pragma unroll
    for (int k = 0; k <= G; ++k) Hs[k] = row16_sum(Hs[k]);
    POL_MARK(8);
    const float r_inv = 1.0f / (sqrtf(Hs[G]) / sqrtf((float)NC) + 1e-6f);
    const float a0 = a.alpha[s][0], a1 = a.alpha[s][1], a2 = a.alpha[s][2];
    const float* __restrict__ be = a.beta[s];
    float pre[2], post[2], A[2][2], u[2] = {1.0f, 1.0f}, v[2] = {1.0f, 1.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pre[i] = sigmoidf_(r_inv * Hs[i] * a0 + be[i]);
      post[i] = 2.0f * sigmoidf_(r_inv * Hs[2 + i] * a1 + be[2 + i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) A[i][j] = exp_(r_inv * Hs[4 + 2 * i + j] * a2 + be[4 + 2 * i + j]);
    }
    for (int it = 0; it < a.sk_it; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i) u[i] = rcp_(A[i][0] * v[0] + A[i][1] * v[1] + 1e-8f);
#pragma unroll
      for (int j = 0; j < 2; ++j) v[j] = rcp_(A[0][j] * u[0] + A[1][j] * u[1] + 1e-8f);
    }
    float mix[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mix[i][j] = u[i] * A[i][j] * v[j];
    POL_MARK(9);
    // read = pre_0 h_0 + pre_1 h_1 -> the Linear's input rows
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 rd;
#pragma unroll
      for (int e = 0; e < 4; ++e) rd[e] = pre[0] * x[q][e] + pre[1] * x[q + 2][e];
      *reinterpret_cast<f32x4*>(&rbuf[lrow][64 * q + 4 * sub]) = rd;
    }
    POL_MARK(10);
    __syncthreads();
    POL_MARK(1);
    // out = SiLU(read W^T + b): this wave's two 16-column tiles
    {
      f32x4 av[8], acc[2] = {zero, zero};
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const f32x4*>(&rbuf[r][16 * j + 4 * qq]);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], wv[t][j][e], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int col = 16 * (2 * wave + t) + r;
        const float bv = a.lb[s][col];
#pragma unroll
        for (int g = 0; g < 4; ++g) obuf[4 * qq + g][col] = silu_(acc[t][g] + bv);
      }
    }
    __syncthreads();
    POL_MARK(2);
    // h'_i = post_i out + mix_i0 h_0 + mix_i1 h_1 (:165), back in the row view
    {
      f32x4 o[2], nx[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) o[q] = *reinterpret_cast<const f32x4*>(&obuf[lrow][64 * q + 4 * sub]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) nx[2 * i + q][e] = post[i] * o[q][e] + (mix[i][0] * x[q][e] + mix[i][1] * x[q + 2][e]);
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = nx[q];
    }
    POL_MARK(3);
  }

  // the heads' first weight tile is requested before the final norm (wave = (head wave / 2, column half wave % 2))
  const float* __restrict__ W1 = a.h1_w[wave >> 1] + (size_t)(128 * (wave & 1) + r) * D + 4 * qq;
  // (image: this wave's eight tiles of its head, tile t at + 512 t)
  const f32x4* __restrict__ W1i = a.img ? reinterpret_cast<const f32x4*>(a.img + (size_t)a.n_sub * kPolSubFloats + (size_t)(wave >> 1) * kPolHeadFloats) +
                                              (size_t)(8 * (wave & 1)) * 512 + lane
                                        : nullptr;
  auto head_tile_load = [&](f32x4 (&w)[8], int t) {
    if (W1i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = W1i[(size_t)t * 512 + 64 * j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const f32x4*>(W1 + (size_t)(16 * t) * D + 16 * j);
    }
  };
  f32x4 wA[8], wB[8];
  head_tile_load(wA, 0);
  // final_norm(h.sum(1)) (:182-183) -> the heads' input rows
  {
    f32x4 sv[2];
    float sq = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) { sv[q][e] = x[q][e] + x[q + 2][e]; sq += sv[q][e] * sv[q][e]; }
    sq = row16_sum(sq);
    const float rr = rsqrtf(sq / (float)D + a.fn_eps);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(a.fn_w + 64 * q + 4 * sub);
      f32x4 f;
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = sv[q][e] * rr * w[e];
      *reinterpret_cast<f32x4*>(&rbuf[lrow][64 * q + 4 * sub]) = f;
    }
  }
  __syncthreads();
  POL_MARK(4);
  // heads (:371-402): wave = (head wave / 2, column half wave % 2) of Linear(128, 256) -> SiLU; the RMSNorm's scale is a per-row
  // factor of the last Linear, so each wave hands over its half's sum of squares and norm-weighted dot products
  {
    const int hd = wave >> 1, half = wave & 1;
    const int n_out = hd == 0 ? a.n_act : 1;
    f32x4 av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const f32x4*>(&rbuf[r][16 * j + 4 * qq]);
    float sq[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dot[4][kPolMaxOut];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int o = 0; o < kPolMaxOut; ++o) dot[g][o] = 0.0f;
    const float* __restrict__ W2 = a.h2_w[hd];
    auto tile = [&](const f32x4* wv, int t) {              // one 16-column tile: MFMAs, SiLU, this tile's share of the sums
      const int col = 128 * half + 16 * t + r;             // column of this head's hidden layer
      f32x4 acc = zero;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], wv[j][e], acc, 0, 0, 0);
      const float bv = a.h1_b[hd][col], nwv = a.hn_w[hd][col];
      float w2[kPolMaxOut];
#pragma unroll
      for (int o = 0; o < kPolMaxOut; ++o) w2[o] = o < n_out ? nwv * W2[(size_t)o * H + col] : 0.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float sv = silu_(acc[g] + bv);
        sq[g] += sv * sv;
#pragma unroll
        for (int o = 0; o < kPolMaxOut; ++o) dot[g][o] += sv * w2[o];
      }
    };
    for (int t = 0; t < 8; t += 2) {                       // the next tile's weights are in flight while this one multiplies
      head_tile_load(wB, t + 1);
      tile(wA, t);
      if (t + 2 < 8) head_tile_load(wA, t + 2);
      tile(wB, t + 1);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      sq[g] = row16_sum(sq[g]);
#pragma unroll
      for (int o = 0; o < kPolMaxOut; ++o) dot[g][o] = row16_sum(dot[g][o]);
    }
    if (r == 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        psq[wave][4 * qq + g] = sq[g];
#pragma unroll
        for (int o = 0; o < kPolMaxOut; ++o) pdot[wave][4 * qq + g][o] = dot[g][o];
      }
    }
  }
  __syncthreads();
  POL_MARK(5);
  if (threadIdx.x < 32) {                                  // thread = (head, row)
    const int hd = threadIdx.x >> 4, lr = threadIdx.x & 15;
    if (lr < n_valid) {
      const float rr = rsqrtf((psq[2 * hd][lr] + psq[2 * hd + 1][lr]) / (float)H + a.hn_eps[hd]);
      const int n_out = hd == 0 ? a.n_act : 1;
      for (int o = 0; o < n_out; ++o) {
        const float y = (pdot[2 * hd][lr][o] + pdot[2 * hd + 1][lr][o]) * rr + a.h2_b[hd][o];
        if (hd == 0) logits_out[lr * ld_logits + o] = y;
        else value_out[lr * ld_value] = y;
      }
    }
  }
  POL_MARK(6);
}

// host side: gymrl_mhc_policy (include/gymrl.h) -> PolicyArgs; -22 when a pointer is missing or the shape is not the kernel's
inline int policy_fill(PolicyArgs& a, const gymrl_mhc_policy* p) {
  if (!p || p->obs_dim < 1 || p->obs_dim > 16 || p->n_sub < 0 || p->n_sub > kPolMaxSub || p->n_act < 1 || p->n_act > kPolMaxOut ||
      p->sk_it < 0 || !p->in_w || !p->in_b || !p->final_norm_w)
    return -22;
  for (int s = 0; s < p->n_sub; ++s) {
    const gymrl_mhc_sub& sb = p->sub[s];
    if (!sb.norm_w || !sb.w || !sb.alpha || !sb.beta || !sb.lin_w || !sb.lin_b) return -22;
    a.norm_w[s] = sb.norm_w; a.gw[s] = sb.w; a.alpha[s] = sb.alpha; a.beta[s] = sb.beta; a.lw[s] = sb.lin_w; a.lb[s] = sb.lin_b;
  }
  for (int h = 0; h < 2; ++h) {
    const gymrl_mhc_head& hd = p->head[h];
    if (!hd.w1 || !hd.b1 || !hd.norm_w || !hd.w2 || !hd.b2) return -22;
    a.h1_w[h] = hd.w1; a.h1_b[h] = hd.b1; a.hn_w[h] = hd.norm_w; a.h2_w[h] = hd.w2; a.h2_b[h] = hd.b2; a.hn_eps[h] = hd.norm_eps;
  }
  a.in_w = p->in_w; a.in_b = p->in_b; a.fn_w = p->final_norm_w; a.fn_eps = p->final_norm_eps;
  a.obs_dim = p->obs_dim; a.n_sub = p->n_sub; a.n_act = p->n_act; a.sk_it = p->sk_it;
  if (reinterpret_cast<uintptr_t>(p->image) & 15) return -22;
  a.img = p->image;
  return 0;
}

}  // namespace mhc
}  // namespace gymrl
