// mhc.hip — PPO-full's manifold-hyper-connection backbone: the rollout forward and the pieces of the training pass.
//
// ppo_full_lunarlander.py:106-250: every MHCBlock half reads the branch stack h [B, n, D] through per-sample gates —
// an RMS-fused linear read-out (n*D -> n*n + 2n numbers per row), two sigmoids, an exp and `max_sk_it` Sinkhorn-Knopp
// sweeps on an n x n matrix — mixes the branches, runs ONE D x D Linear + SiLU on the weighted branch sum and writes
// the stack back.  Through PyTorch that is ~95 launches per half (each Sinkhorn sweep alone is 6), ~400 per rollout
// forward at 4096 rows: 3 ms per vector step of pure launch cost, 12 of the 47 s of a config-5 iteration.
//
// Rollout (no gradients):
//   gymrl_mhc_policy_forward   the whole ActorCritic.forward in one launch for the default shape (n = 2, D = 128, 256-wide heads)
//   per layer, any other shape: gymrl_mhc_gates (gates + read = sum_i pre_i h_i), gymrl_lin_fwd (csrc/lin.hip), gymrl_mhc_combine
//   (h'[b, i, :] = post_i out + sum_j mix_ij h[b, j, :]), gymrl_rmsnorm (optionally over the branch sum / of SiLU(x))
// Training pass (autograd nodes in gymrl_amd/ppo_full_lunarlander.py: _MhcSub, _RmsNorm; _MhcGates / _MhcRead / _MhcCombine):
//   gymrl_mhc_gates (+ stats) / gymrl_mhc_gates_bwd, gymrl_mhc_combine(_bwd) with SiLU on load, gymrl_mhc_read_fwd/_bwd,
//   gymrl_rmsnorm / gymrl_rmsnorm_bwd, gymrl_sinkhorn; parameter gradients are per-workgroup partial sums added in a fixed order.
// All floating point, compared with the torch modules in float64 at 1e-5 (gradients 2-3e-5): tests/test_mhc_fused_gpu.py.
#include "mhc_policy_device.hpp"

namespace {

using namespace gymrl;
using namespace gymrl::mhc;

constexpr int kWaves = 4;

struct GatesArgs {
  const float* h; const float* norm_w; const float* w; const float* alpha; const float* beta;
  float* pre; float* post; float* mix; float* read; float* stats;
  int B, D, sk_it;
};

template <int N>
__global__ __launch_bounds__(64 * kWaves) void mhc_gates_kernel(const GatesArgs a) {
  constexpr int G = N * N + 2 * N;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= a.B) return;
  const int nc = N * a.D;
  const float* __restrict__ hr = a.h + (size_t)row * nc;
  float Hs[G], sq = 0.0f;
#pragma unroll
  for (int j = 0; j < G; ++j) Hs[j] = 0.0f;
  for (int c = 4 * lane; c < nc; c += 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(hr + c);
    const f32x4 nw = *reinterpret_cast<const f32x4*>(a.norm_w + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = nw[e] * x[e];
      sq += x[e] * x[e];
      const float* wr = a.w + (size_t)(c + e) * G;
#pragma unroll
      for (int j = 0; j < G; ++j) Hs[j] += t * wr[j];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sq += __shfl_xor(sq, off, 64);
#pragma unroll
    for (int j = 0; j < G; ++j) Hs[j] += __shfl_xor(Hs[j], off, 64);
  }
  // every lane now holds the row's sums: r_inv = 1 / (|flat| / sqrt(nc) + 1e-6)
  const float r_inv = 1.0f / (sqrtf(sq) / sqrtf((float)nc) + 1e-6f);
  const float a0 = a.alpha[0], a1 = a.alpha[1], a2 = a.alpha[2];
  float pre[N], post[N], A[N][N], u[N], v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    pre[i] = sigmoidf_(r_inv * Hs[i] * a0 + a.beta[i]);
    post[i] = 2.0f * sigmoidf_(r_inv * Hs[N + i] * a1 + a.beta[N + i]);
    u[i] = 1.0f; v[i] = 1.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) A[i][j] = exp_(r_inv * Hs[2 * N + i * N + j] * a2 + a.beta[2 * N + i * N + j]);
  }
  for (int it = 0; it < a.sk_it; ++it) {                   // Sinkhorn-Knopp scalings (:141-146)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < N; ++j) s += A[i][j] * v[j];
      u[i] = rcp_(s + 1e-8f);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) s += A[i][j] * u[i];
      v[j] = rcp_(s + 1e-8f);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      a.pre[(size_t)row * N + i] = pre[i];
      a.post[(size_t)row * N + i] = post[i];
#pragma unroll
      for (int j = 0; j < N; ++j) a.mix[((size_t)row * N + i) * N + j] = u[i] * A[i][j] * v[j];
    }
  }
  for (int d = lane; d < a.D; d += 64) {                   // read = bmm(pre, h): the weighted sum of the branches
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) s += pre[i] * hr[i * a.D + d];
    a.read[(size_t)row * a.D + d] = s;
  }
}

// The n = 2 gates at nc = 256 * CH columns.  The one-wave-per-row kernel above spends ~1100 of its ~1500 instructions per row on
// the Sinkhorn sweeps, every lane repeating them (0.3 ms at 131072 rows against 50 us of HBM time; 23 us per rollout call at 4096
// rows).  Here a wave takes RB = 16 / CH rows: 16 lanes per row (a 256-byte segment per load, all of the batch's loads issued up
// front and kept in registers for the read-out), the read-out sums through DPP, then ONE lane per row does the sigmoids, the exp
// and the sweeps, and the branch sum is formed from the registers.  stats [B, 9] (optional) = the eight read-out sums and
// |flat|^2 of the row, for gymrl_mhc_gates_bwd.
template <int CH>
__global__ __launch_bounds__(64) void mhc_gates2_kernel(const GatesArgs a) {
  constexpr int N = 2, G = 8, IT = 4 / CH, RB = 4 * IT, Q = 4 * CH;
  const int lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
  const int nc = 256 * CH;
  const int64_t base = (int64_t)blockIdx.x * RB;
  f32x4 x[IT][Q];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    int64_t row = base + grp * IT + it;
    if (row > a.B - 1) row = a.B - 1;
    const float* hr = a.h + row * nc + 4 * sub;
#pragma unroll
    for (int q = 0; q < Q; ++q) x[it][q] = *reinterpret_cast<const f32x4*>(hr + 64 * q);
  }
  float Hs[IT][G + 1];
#pragma unroll
  for (int it = 0; it < IT; ++it)
#pragma unroll
    for (int k = 0; k <= G; ++k) Hs[it][k] = 0.0f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = 64 * q + 4 * sub;
    const f32x4 nw = *reinterpret_cast<const f32x4*>(a.norm_w + c);
    float wq[4][G];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(a.w + (size_t)(c + e) * G);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(a.w + (size_t)(c + e) * G + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { wq[e][k] = lo[k]; wq[e][4 + k] = hi[k]; }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xv = x[it][q][e], t = nw[e] * xv;
        Hs[it][G] += xv * xv;
#pragma unroll
        for (int k = 0; k < G; ++k) Hs[it][k] += t * wq[e][k];
      }
  }
  float mine[G + 1];                                       // lane `sub` of a group keeps the sums of the group's row `sub`
#pragma unroll
  for (int k = 0; k <= G; ++k) mine[k] = 0.0f;
#pragma unroll
  for (int it = 0; it < IT; ++it)
#pragma unroll
    for (int k = 0; k <= G; ++k) {
      const float s = row16_sum(Hs[it][k]);
      mine[k] = sub == it ? s : mine[k];
    }
  const int64_t my_row = base + grp * IT + sub;
  float pre[N] = {0.0f, 0.0f};
  if (sub < IT && my_row < a.B) {
    const float r_inv = 1.0f / (sqrtf(mine[G]) / sqrtf((float)nc) + 1e-6f);
    const float a0 = a.alpha[0], a1 = a.alpha[1], a2 = a.alpha[2];
    float post[N], A[N][N], u[N], v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      pre[i] = sigmoidf_(r_inv * mine[i] * a0 + a.beta[i]);
      post[i] = 2.0f * sigmoidf_(r_inv * mine[N + i] * a1 + a.beta[N + i]);
      u[i] = 1.0f; v[i] = 1.0f;
#pragma unroll
      for (int j = 0; j < N; ++j) A[i][j] = exp_(r_inv * mine[2 * N + i * N + j] * a2 + a.beta[2 * N + i * N + j]);
    }
    for (int it = 0; it < a.sk_it; ++it) {                 // Sinkhorn-Knopp scalings (:141-146)
#pragma unroll
      for (int i = 0; i < N; ++i) u[i] = rcp_(A[i][0] * v[0] + A[i][1] * v[1] + 1e-8f);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = rcp_(A[0][j] * u[0] + A[1][j] * u[1] + 1e-8f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      a.pre[my_row * N + i] = pre[i];
      a.post[my_row * N + i] = post[i];
#pragma unroll
      for (int j = 0; j < N; ++j) a.mix[(my_row * N + i) * N + j] = u[i] * A[i][j] * v[j];
    }
    if (a.stats) {
#pragma unroll
      for (int k = 0; k <= G; ++k) a.stats[my_row * (G + 1) + k] = mine[k];
    }
  }
  // read = pre_0 h_0 + pre_1 h_1 from the registers: row (grp, it)'s gates live in lane 16 grp + it
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int src = ((lane & 48) + it) << 2;
    const float p0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pre[0])));
    const float p1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pre[1])));
    const int64_t row = base + grp * IT + it;
    if (row < a.B) {
#pragma unroll
      for (int q = 0; q < Q / 2; ++q) {
        f32x4 s;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = p0 * x[it][q][e] + p1 * x[it][q + Q / 2][e];
        *reinterpret_cast<f32x4*>(a.read + row * (nc / 2) + 64 * q + 4 * sub) = s;
      }
    }
  }
}

template <int N>
__global__ __launch_bounds__(256) void mhc_combine_kernel(const float* __restrict__ post, const float* __restrict__ mix,
                                                          const float* __restrict__ out, const float* __restrict__ h, int B,
                                                          int D, int silu, float* __restrict__ h_out) {
  const int64_t total = (int64_t)B * D;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t b = t / D;
    const int d = (int)(t % D);
    float hv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) hv[j] = h[(b * N + j) * D + d];
    float o = out[b * D + d];
    if (silu) o = silu_(o);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < N; ++j) s += mix[(b * N + i) * N + j] * hv[j];
      h_out[(b * N + i) * D + d] = post[b * N + i] * o + s;
    }
  }
}

// ---- training pass: the two branch-mixing products of a hyper-connection with their backward, one launch each way ----
// read[b, :] = sum_i pre[b, i] h[b, i, :]                        (MHCBlock._sub :161)
template <int N>
__global__ __launch_bounds__(256) void mhc_read_fwd_kernel(const float* __restrict__ pre, const float* __restrict__ h, int B, int D,
                                                           float* __restrict__ read) {
  const int64_t total = (int64_t)B * (D >> 2);
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t b = t / (D >> 2);
    const int d = (int)(t % (D >> 2)) * 4;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float p = pre[b * N + i];
      const f32x4 x = *reinterpret_cast<const f32x4*>(h + (b * N + i) * D + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += p * x[e];
    }
    *reinterpret_cast<f32x4*>(read + b * D + d) = s;
  }
}

// one wave per row: d_pre[b, i] = sum_d g[b, d] h[b, i, d];  d_h[b, i, d] (+)= pre[b, i] g[b, d]  (d_h == nullptr: d_pre only)
template <int N>
__global__ __launch_bounds__(64 * kWaves) void mhc_read_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pre,
                                                                 const float* __restrict__ h, int B, int D,
                                                                 float* __restrict__ d_pre, float* __restrict__ d_h, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= B) return;
  float p[N], acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { p[i] = pre[row * N + i]; acc[i] = 0.0f; }
  for (int d = lane; d < D; d += 64) {
    const float gv = g[row * D + d];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int64_t o = (row * N + i) * D + d;
      acc[i] += gv * h[o];
      if (d_h) d_h[o] = accumulate ? d_h[o] + p[i] * gv : p[i] * gv;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
    if (lane == 0) d_pre[row * N + i] = acc[i];
  }
}

// backward of h'[b, i, :] = post[b, i] out[b, :] + sum_j mix[b, i, j] h[b, j, :], one wave per row:
//   d_post[i] = sum_d g[i, d] out[d];  d_out[d] = sum_i post[i] g[i, d];  d_mix[i, j] = sum_d g[i, d] h[j, d];  d_h[j, d] = sum_i mix[i, j] g[i, d]
template <int N>
__global__ __launch_bounds__(64 * kWaves) void mhc_combine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ post,
                                                                    const float* __restrict__ mix, const float* __restrict__ out,
                                                                    const float* __restrict__ h, int B, int D, int silu,
                                                                    float* __restrict__ d_post, float* __restrict__ d_mix,
                                                                    float* __restrict__ d_out, float* __restrict__ d_h) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= B) return;
  float po[N], mx[N][N], a_post[N], a_mix[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    po[i] = post[row * N + i]; a_post[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) { mx[i][j] = mix[(row * N + i) * N + j]; a_mix[i][j] = 0.0f; }
  }
  for (int d = lane; d < D; d += 64) {
    float gv[N], hv[N];
    const float z = out[row * D + d];
    const float o = silu ? silu_(z) : z;
#pragma unroll
    for (int i = 0; i < N; ++i) { gv[i] = g[(row * N + i) * D + d]; hv[i] = h[(row * N + i) * D + d]; }
    float so = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      so += po[i] * gv[i];
      a_post[i] += gv[i] * o;
#pragma unroll
      for (int j = 0; j < N; ++j) a_mix[i][j] += gv[i] * hv[j];
    }
    d_out[row * D + d] = silu ? so * silu_grad_(z) : so;        // silu: `out` holds z and d_out is dL/dz
    if (d_h) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float sh = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) sh += mx[i][j] * gv[i];
        d_h[(row * N + j) * D + d] = sh;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a_post[i] += __shfl_xor(a_post[i], off, 64);
    if (lane == 0) d_post[row * N + i] = a_post[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a_mix[i][j] += __shfl_xor(a_mix[i][j], off, 64);
      if (lane == 0) d_mix[(row * N + i) * N + j] = a_mix[i][j];
    }
  }
}

// ---- training pass: backward of the gates (n = 2 branches) -----------------------------------------------------------
// z = r H alpha + beta with H = (norm_w * flat) w, r = 1 / (|flat| / sqrt(nc) + 1e-6);  pre = sigmoid(z[:n]), post = 2 sigmoid(z[n:2n]),
// mix = u exp(z[2n:]) v with u, v constants (the reference computes them under no_grad).  The forward saved H and |flat|^2 per row
// (stats), so nothing here needs a reduction over a row's columns:
//   phase A, one LANE per row, 64 rows per wave step: dz (sigmoid' / exp' from the saved outputs), dH = dz r alpha,
//            d|flat| / |flat| from d r = sum dz H alpha, and the row's terms of d alpha, d beta;
//   phase B, one lane per 4 columns (a wave covers 256; blockIdx.y picks the 256-column block when nc = 512), streaming the
//            wave's rows two at a time with row r's nine scalars read from lane r (v_readlane -> SGPRs):
//            d flat = norm_w (dH w^T) + d|flat| flat / |flat|   [+ pre_j d_read + sum_i mix_ij g_i: the sub-block's other two
//            consumers of h, folded in so that autograd has nothing to add], and the columns' terms of d norm_w, d w in registers.
// The first version recomputed H with 54 ds_bpermute per row at 2 waves per SIMD and ran 0.44 ms at 131072 rows (0.6 TB/s).
// Parameter gradients: added across the workgroup's waves through LDS in a fixed order, one partial vector per workgroup,
// summed ascending by partial_reduce_kernel: no atomics.
struct GatesBwdArgs {
  const float* h; const float* norm_w; const float* w; const float* alpha;
  const float* pre; const float* post; const float* mix; const float* stats;
  const float* d_pre; const float* d_post; const float* d_mix;
  const float* d_read;                                     // nullable [B, D]: d_h[b, j] += pre[b, j] d_read[b]
  const float* g_out;                                      // nullable [B, 2, D]: d_h[b, j] += sum_i mix[b, i, j] g_out[b, i]
  float* d_h; float* partial;
  int B, D;
};

constexpr int kGatesLen = 256 + 256 * 8 + 3 + 8;           // one 256-column block's partial: d norm_w, d w, d alpha, d beta

__device__ __forceinline__ float lane_value(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}

__global__ __launch_bounds__(64 * kWaves) void mhc_gates_bwd_kernel(const GatesBwdArgs a) {
  constexpr int N = 2, G = N * N + 2 * N, U = 2;
  extern __shared__ float red[];                           // [kWaves][kGatesLen]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = N * a.D;
  const int c0 = 256 * blockIdx.y + 4 * lane;              // this lane's four columns of flat
  const int j = c0 / a.D, d0 = c0 % a.D;                   // = branch j, columns d0 .. d0 + 3
  const f32x4 nw = *reinterpret_cast<const f32x4*>(a.norm_w + c0);
  float wr[4][G];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(a.w + (size_t)(c0 + e) * G);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(a.w + (size_t)(c0 + e) * G + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { wr[e][k] = lo[k]; wr[e][4 + k] = hi[k]; }
  }
  const float al[3] = {a.alpha[0], a.alpha[1], a.alpha[2]};
  float acc_nw[4], acc_w[4][G], acc_al[3] = {0.0f, 0.0f, 0.0f}, acc_be[G];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc_nw[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < G; ++k) acc_w[e][k] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < G; ++k) acc_be[k] = 0.0f;
  const float inv_sqrt_nc = 1.0f / sqrtf((float)nc);
  for (int64_t base = ((int64_t)blockIdx.x * kWaves + wave) * 64; base < a.B; base += (int64_t)gridDim.x * kWaves * 64) {
    // ---- phase A: lane = row
    const int64_t row = base + lane;
    float dH[G], dn_over = 0.0f, p0 = 0.0f, p1 = 0.0f, m00 = 0.0f, m01 = 0.0f, m10 = 0.0f, m11 = 0.0f;
#pragma unroll
    for (int k = 0; k < G; ++k) dH[k] = 0.0f;
    if (row < a.B) {
      float Hs[G], dz[G];
#pragma unroll
      for (int k = 0; k < G; ++k) Hs[k] = a.stats[row * (G + 1) + k];
      const float norm = sqrtf(a.stats[row * (G + 1) + G]);
      const float r = 1.0f / (norm * inv_sqrt_nc + 1e-6f);
      p0 = a.pre[row * N]; p1 = a.pre[row * N + 1];
      const float q0 = a.post[row * N], q1 = a.post[row * N + 1];
      m00 = a.mix[row * 4]; m01 = a.mix[row * 4 + 1]; m10 = a.mix[row * 4 + 2]; m11 = a.mix[row * 4 + 3];
      dz[0] = a.d_pre[row * N] * p0 * (1.0f - p0);
      dz[1] = a.d_pre[row * N + 1] * p1 * (1.0f - p1);
      dz[2] = a.d_post[row * N] * q0 * (1.0f - 0.5f * q0);
      dz[3] = a.d_post[row * N + 1] * q1 * (1.0f - 0.5f * q1);
      dz[4] = a.d_mix[row * 4] * m00; dz[5] = a.d_mix[row * 4 + 1] * m01;
      dz[6] = a.d_mix[row * 4 + 2] * m10; dz[7] = a.d_mix[row * 4 + 3] * m11;
      float d_r = 0.0f;
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int gi = k < N ? 0 : (k < 2 * N ? 1 : 2);
        dH[k] = dz[k] * r * al[gi];
        d_r += dz[k] * Hs[k] * al[gi];
        acc_al[gi] += dz[k] * r * Hs[k];
        acc_be[k] += dz[k];
      }
      const float d_norm = d_r * (-r * r * inv_sqrt_nc);
      dn_over = norm > 0.0f ? d_norm / norm : 0.0f;
    }
    // ---- phase B: lane = 4 columns, the wave's rows in pairs
    const int nrows = (int)((a.B - base) < 64 ? (a.B - base) : 64);
    for (int r0 = 0; r0 < nrows; r0 += U) {
      f32x4 x[U], gr[U], g0[U], g1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r0 + u < nrows ? r0 + u : nrows - 1;
        const int64_t rw = base + rr;
        x[u] = *reinterpret_cast<const f32x4*>(a.h + rw * nc + c0);
        if (a.d_read) gr[u] = *reinterpret_cast<const f32x4*>(a.d_read + rw * a.D + d0);
        if (a.g_out) {
          g0[u] = *reinterpret_cast<const f32x4*>(a.g_out + (rw * N) * a.D + d0);
          g1[u] = *reinterpret_cast<const f32x4*>(a.g_out + (rw * N + 1) * a.D + d0);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r0 + u;
        if (rr < nrows) {
          float sH[G];
#pragma unroll
          for (int k = 0; k < G; ++k) sH[k] = lane_value(dH[k], rr);
          const float s_dn = lane_value(dn_over, rr);
          f32x4 dx;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t2 = 0.0f;
#pragma unroll
            for (int k = 0; k < G; ++k) t2 += sH[k] * wr[e][k];
            dx[e] = nw[e] * t2 + s_dn * x[u][e];
            acc_nw[e] += x[u][e] * t2;
            const float t = nw[e] * x[u][e];
#pragma unroll
            for (int k = 0; k < G; ++k) acc_w[e][k] += t * sH[k];
          }
          if (a.d_read) {
            const float pj = j ? lane_value(p1, rr) : lane_value(p0, rr);
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] += pj * gr[u][e];
          }
          if (a.g_out) {
            const float m0j = j ? lane_value(m01, rr) : lane_value(m00, rr);
            const float m1j = j ? lane_value(m11, rr) : lane_value(m10, rr);
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] += m0j * g0[u][e] + m1j * g1[u][e];
          }
          *reinterpret_cast<f32x4*>(a.d_h + (base + rr) * nc + c0) = dx;
        }
      }
    }
  }
  // d alpha / d beta: the lanes' row sums added across the wave (fixed tree), then everything across the workgroup's waves
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int g = 0; g < 3; ++g) acc_al[g] += __shfl_xor(acc_al[g], off, 64);
#pragma unroll
    for (int k = 0; k < G; ++k) acc_be[k] += __shfl_xor(acc_be[k], off, 64);
  }
  float* mine = red + (size_t)wave * kGatesLen;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * lane + e;
    mine[c] = acc_nw[e];
#pragma unroll
    for (int k = 0; k < G; ++k) mine[256 + c * G + k] = acc_w[e][k];
  }
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) mine[256 + 256 * G + g] = acc_al[g];
#pragma unroll
    for (int k = 0; k < G; ++k) mine[256 + 256 * G + 3 + k] = acc_be[k];
  }
  __syncthreads();
  float* out = a.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kGatesLen;
  for (int i = threadIdx.x; i < kGatesLen; i += 64 * kWaves) {
    float sum = red[i];
#pragma unroll
    for (int w2 = 1; w2 < kWaves; ++w2) sum += red[(size_t)w2 * kGatesLen + i];
    out[i] = sum;
  }
}

// out[i] = sum over b < blocks (ascending within eight fixed slices, the slices ascending) of partial[(y * blocks + b) * len + i];
// the destination of element i of column block y is the segment it falls in: seg_end[s - 1] <= i < seg_end[s] ->
// dst[s][y * seg_stride[s] + i - seg_end[s - 1]]  (seg_stride 0: only column block 0 writes the segment)
struct ReduceArgs {
  const float* partial; int blocks, len, n_seg;
  int seg_end[4]; int seg_stride[4]; float* dst[4];
};

__global__ __launch_bounds__(256) void partial_reduce_kernel(const ReduceArgs a) {
  __shared__ float part[8][32];
  const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col, y = blockIdx.y;
  float s = 0.0f;
  if (i < a.len) {
    const int per = (a.blocks + 7) / 8;
    const int b0 = sl * per, b1 = b0 + per < a.blocks ? b0 + per : a.blocks;
    const float* p = a.partial + (size_t)y * a.blocks * a.len + i;
#pragma unroll 8                                           // eight loads in flight (one at a time: 64 us for 512 partials)
    for (int b = b0; b < b1; ++b) s += p[(size_t)b * a.len];
  }
  part[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < a.len) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[k][col];
    int lo = 0;
    for (int sg = 0; sg < a.n_seg; ++sg) {
      if (i < a.seg_end[sg]) {
        if (a.seg_stride[sg] || y == 0) a.dst[sg][(size_t)y * a.seg_stride[sg] + (i - lo)] = s;
        break;
      }
      lo = a.seg_end[sg];
    }
  }
}

// Sinkhorn-Knopp scalings of B positive n x n matrices (ManifoldHyperConnectionFuse.gates :141-146, under no_grad in the
// reference: u, v are constants of the backward pass): one lane per matrix instead of ~6 launches per sweep.
template <int N>
__global__ __launch_bounds__(256) void sinkhorn_kernel(const float* __restrict__ A, int B, int sk_it, float* __restrict__ u_out,
                                                       float* __restrict__ v_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float a[N][N], u[N], v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u[i] = 1.0f; v[i] = 1.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) a[i][j] = A[((size_t)b * N + i) * N + j];
  }
  for (int it = 0; it < sk_it; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < N; ++j) s += a[i][j] * v[j];
      u[i] = rcp_(s + 1e-8f);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) s += a[i][j] * u[i];
      v[j] = rcp_(s + 1e-8f);
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { u_out[(size_t)b * N + i] = u[i]; v_out[(size_t)b * N + i] = v[i]; }
}

// y = s * rsqrt(mean(s^2) + eps) * w per row; n_sum > 1: s = the sum of n_sum consecutive [D] blocks of the row;
// silu: s = SiLU(x) (the MLPs' Linear -> SiLU -> RMSNorm: the activation rides in the norm's two launches)
__global__ __launch_bounds__(64 * kWaves) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             int B, int D, int n_sum, float eps, int silu, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + (size_t)row * n_sum * D;
  float sq = 0.0f;
  for (int d = lane; d < D; d += 64) {
    float s = xr[d];
    for (int k = 1; k < n_sum; ++k) s += xr[k * D + d];
    if (silu) s = silu_(s);
    sq += s * s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  const float r = rsqrtf(sq / (float)D + eps);
  for (int d = lane; d < D; d += 64) {
    float s = xr[d];
    for (int k = 1; k < n_sum; ++k) s += xr[k * D + d];
    if (silu) s = silu_(s);
    y[(size_t)row * D + d] = s * r * w[d];
  }
}

// the same with the row in registers (D <= 64 Q): one read of x, SiLU evaluated once
template <int Q>
__global__ __launch_bounds__(64 * kWaves) void rmsnorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 int B, int D, int n_sum, float eps, int silu, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  float wv[Q];                                             // the norm's weight: constants of the launch
#pragma unroll
  for (int q = 0; q < Q; ++q) wv[q] = lane + 64 * q < D ? w[lane + 64 * q] : 0.0f;
  // a wave walks rows (262144 one-row waves cost more in dispatch than in HBM time: 2.8 TB/s)
  for (int64_t row = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); row < B; row += (int64_t)gridDim.x * kWaves) {
    const float* xr = x + (size_t)row * n_sum * D;
    float sv[Q], sq = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int d = lane + 64 * q;
      float s = 0.0f;
      if (d < D) {
        s = xr[d];
        for (int k = 1; k < n_sum; ++k) s += xr[k * D + d];
        if (silu) s = silu_(s);
      }
      sv[q] = s;
      sq += s * s;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float r = rsqrtf(sq / (float)D + eps);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int d = lane + 64 * q;
      if (d < D) y[(size_t)row * D + d] = sv[q] * r * wv[q];
    }
  }
}

// backward of y = s r w, s = x or SiLU(x), r = rsqrt(mean(s^2) + eps), one wave per row (D <= 512, lane l holds columns l + 64 q):
//   d s = r (w g) - s r^3 / D sum_d(w g s);  d x = d s [SiLU'(x)];  d w[d] = sum over rows g s r — per-lane column sums over the rows
// the wave visits, added across the workgroup's waves through LDS, one partial vector per workgroup for partial_reduce_kernel.
template <int kNormQ>
__global__ __launch_bounds__(64 * kWaves) void rmsnorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 const float* __restrict__ w, int B, int D, int n_sum, float eps, int silu,
                                                                 float* __restrict__ d_x, float* __restrict__ partial) {
  __shared__ float red[kWaves][64 * kNormQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[kNormQ], acc[kNormQ];
#pragma unroll
  for (int q = 0; q < kNormQ; ++q) {
    const int d = lane + 64 * q;
    wv[q] = d < D ? w[d] : 0.0f;
    acc[q] = 0.0f;
  }
  const float inv_d = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < B; row += (int64_t)gridDim.x * kWaves) {
    float xv[kNormQ], gv[kNormQ], sv[kNormQ], sq = 0.0f, dot = 0.0f;
#pragma unroll
    for (int q = 0; q < kNormQ; ++q) {
      const int d = lane + 64 * q;
      float xs = 0.0f;                                     // x = the sum of the row's n_sum blocks (ascending, as the forward adds them)
      if (d < D)
        for (int i = 0; i < n_sum; ++i) xs += x[(row * n_sum + i) * D + d];
      xv[q] = xs;
      gv[q] = d < D ? g[row * D + d] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kNormQ; ++q) {
      sv[q] = silu ? silu_(xv[q]) : xv[q];
      sq += sv[q] * sv[q];
      dot += wv[q] * gv[q] * sv[q];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sq += __shfl_xor(sq, off, 64);
      dot += __shfl_xor(dot, off, 64);
    }
    const float r = rsqrtf(sq * inv_d + eps);
    const float k3 = r * r * r * inv_d * dot;
#pragma unroll
    for (int q = 0; q < kNormQ; ++q) {
      const int d = lane + 64 * q;
      if (d < D) {
        float ds = r * wv[q] * gv[q] - sv[q] * k3;
        if (silu) ds *= silu_grad_(xv[q]);
        d_x[row * D + d] = ds;
        acc[q] += gv[q] * sv[q] * r;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kNormQ; ++q) red[wave][lane + 64 * q] = acc[q];
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 64 * kWaves) {
    float sum = red[0][i];
#pragma unroll
    for (int w2 = 1; w2 < kWaves; ++w2) sum += red[w2][i];
    partial[(size_t)blockIdx.x * D + i] = sum;
  }
}

// ---- training pass: a head's tail, SiLU -> RMSNorm -> Linear(D -> n_out <= 8), in one launch each way ---------------------
// MLP([128, 256, n_out]) (:371-402) ends in y = RMSNorm(SiLU(x)), out = y W2^T + b2 with n_out = 4 (actor) or 1 (critic).  As the
// norm's launches plus the layer kernels that is 1 KB of y per row written, read back twice (the projection, its weight
// gradient) and a [B, D] gradient d y written and re-read: 2.2 KB per row of traffic that carries 16 bytes of information.
// Here a wave walks rows with the row in registers (rmsnorm_reg_kernel's layout: lane l holds columns l + 64 q): forward =
// one read of x, n_out + 1 wave sums; backward = x and the n_out output gradients in, d x out, y recomputed for d W2, and
// the three parameter sums (d norm_w, d W2, d b2) per lane over the rows the wave visits, added across the workgroup through
// LDS: one partial vector per workgroup for partial_reduce_kernel.
template <int Q, int NO>
__global__ __launch_bounds__(64 * kWaves) void norm_proj_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ W2, const float* __restrict__ b2, int B,
                                                                   int D, int n_out, float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  float wv[Q], W2r[NO][Q], b2r[NO];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int d = lane + 64 * q;
    wv[q] = d < D ? w[d] : 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) W2r[o][q] = (o < n_out && d < D) ? W2[(size_t)o * D + d] : 0.0f;
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) b2r[o] = (o < n_out && b2) ? b2[o] : 0.0f;
  for (int64_t row = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); row < B; row += (int64_t)gridDim.x * kWaves) {
    float sq = 0.0f, dot[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) dot[o] = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int d = lane + 64 * q;
      const float sv = d < D ? silu_(x[row * D + d]) : 0.0f;
      sq += sv * sv;
      const float t = sv * wv[q];
#pragma unroll
      for (int o = 0; o < NO; ++o) dot[o] += t * W2r[o][q];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sq += __shfl_xor(sq, off, 64);
#pragma unroll
      for (int o = 0; o < NO; ++o) dot[o] += __shfl_xor(dot[o], off, 64);
    }
    const float r = rsqrtf(sq / (float)D + eps);
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (lane == o && o < n_out) out[row * n_out + o] = r * dot[o] + b2r[o];
  }
}

// The forward at D = 256 with FOUR rows per wave (the sub-block kernels' layout: lane (grp, sub) holds columns 64 q + 4 sub .. + 3 of
// row 4 it + grp): 16-byte loads, a row's n_out + 1 sums are four DPP adds across sixteen lanes instead of six ds_bpermute
// butterflies across sixty-four, and two row quads are in flight per wave.  One row per wave (above) read x at 2.3 (n_out = 4) /
// 3.0 TB/s (n_out = 1) at 524 288 rows: 231 / 176 us per launch of PPO-full's update.
template <int NO>
__global__ __launch_bounds__(64 * kWaves) void norm_proj_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ W2, const float* __restrict__ b2, int B,
                                                                    int n_out, float eps, float* __restrict__ out) {
  constexpr int D = 256;
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  f32x4 wv[4], W2r[NO][4];
  float b2r[NO];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    wv[q] = *reinterpret_cast<const f32x4*>(w + 64 * q + 4 * sub);
#pragma unroll
    for (int o = 0; o < NO; ++o)
      W2r[o][q] = o < n_out ? *reinterpret_cast<const f32x4*>(W2 + (size_t)o * D + 64 * q + 4 * sub) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) b2r[o] = (o < n_out && b2) ? b2[o] : 0.0f;
  const int64_t quads = ((int64_t)B + 3) >> 2;
  const int64_t q0 = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6), qs = (int64_t)gridDim.x * kWaves;
  auto load = [&](f32x4 (&v)[4], int64_t quad) {
    int64_t row = 4 * quad + grp;
    if (row > B - 1) row = B - 1;
    const float* xr = x + row * D + 4 * sub;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(xr + 64 * q);
  };
  f32x4 cur[4], nxt[4];
  if (q0 < quads) load(cur, q0);
  for (int64_t quad = q0; quad < quads; quad += qs) {
    if (quad + qs < quads) load(nxt, quad + qs);
    float sq = 0.0f, dot[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) dot[o] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sv = silu_(cur[q][e]);
        sq += sv * sv;
        const float t = sv * wv[q][e];
#pragma unroll
        for (int o = 0; o < NO; ++o) dot[o] += t * W2r[o][q][e];
      }
    sq = row16_sum(sq);
#pragma unroll
    for (int o = 0; o < NO; ++o) dot[o] = row16_sum(dot[o]);
    const float r = rsqrtf(sq / (float)D + eps);
    const int64_t row = 4 * quad + grp;
    if (row < B) {
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (sub == o && o < n_out) out[row * n_out + o] = r * dot[o] + b2r[o];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
}

template <int Q, int NO>
__global__ __launch_bounds__(64 * kWaves) void norm_proj_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ x,
                                                                   const float* __restrict__ w, const float* __restrict__ W2, int B, int D,
                                                                   int n_out, float eps, float* __restrict__ d_x, float* __restrict__ partial) {
  extern __shared__ float np_red[];                        // [kWaves][len], len = D + n_out * (D + 1)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int len = D + n_out * (D + 1);
  float wv[Q], W2r[NO][Q], acc_w[Q], acc_W2[NO][Q], acc_b[NO];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int d = lane + 64 * q;
    wv[q] = d < D ? w[d] : 0.0f;
    acc_w[q] = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) { W2r[o][q] = (o < n_out && d < D) ? W2[(size_t)o * D + d] : 0.0f; acc_W2[o][q] = 0.0f; }
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) acc_b[o] = 0.0f;
  const float inv_d = 1.0f / (float)D;
  for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < B; row += (int64_t)gridDim.x * kWaves) {
    float xv[Q], sv[Q], gv[Q], dlv[NO], sq = 0.0f, dot = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) dlv[o] = o < n_out ? dl[row * n_out + o] : 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int d = lane + 64 * q;
      xv[q] = d < D ? x[row * D + d] : 0.0f;
      sv[q] = silu_(xv[q]);
      float g = 0.0f;
#pragma unroll
      for (int o = 0; o < NO; ++o) g += dlv[o] * W2r[o][q];
      gv[q] = g;
      sq += sv[q] * sv[q];
      dot += wv[q] * g * sv[q];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sq += __shfl_xor(sq, off, 64);
      dot += __shfl_xor(dot, off, 64);
    }
    const float r = rsqrtf(sq * inv_d + eps);
    const float k3 = r * r * r * inv_d * dot;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int d = lane + 64 * q;
      if (d < D) {
        d_x[row * D + d] = (r * wv[q] * gv[q] - sv[q] * k3) * silu_grad_(xv[q]);
        acc_w[q] += gv[q] * sv[q] * r;
        const float y = sv[q] * r * wv[q];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc_W2[o][q] += dlv[o] * y;
      }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) acc_b[o] += dlv[o];
  }
  float* mine = np_red + (size_t)wave * len;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int d = lane + 64 * q;
    if (d < D) {
      mine[d] = acc_w[q];
#pragma unroll
      for (int o = 0; o < NO; ++o)
        if (o < n_out) mine[D + o * D + d] = acc_W2[o][q];
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (o < n_out) mine[D + n_out * D + o] = acc_b[o];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < len; i += 64 * kWaves) {
    float sum = np_red[i];
#pragma unroll
    for (int w2 = 1; w2 < kWaves; ++w2) sum += np_red[(size_t)w2 * len + i];
    partial[(size_t)blockIdx.x * len + i] = sum;
  }
}

// The backward at D = 256 with four rows per wave (norm_proj_fwd4_kernel's layout): x in 16-byte loads, d x in 16-byte stores, a
// row's two sums four DPP adds; the per-lane parameter sums (d norm_w, d W2, d b2 over the rows the lane sees) are folded across the
// four row groups by two butterflies per accumulator at the END, then across the workgroup's waves through LDS as before.
template <int NO>
__global__ __launch_bounds__(64 * kWaves) void norm_proj_bwd4_kernel(const float* __restrict__ dl, const float* __restrict__ x,
                                                                    const float* __restrict__ w, const float* __restrict__ W2, int B,
                                                                    int n_out, float eps, float* __restrict__ d_x, float* __restrict__ partial) {
  constexpr int D = 256;
  extern __shared__ float np_red[];                        // [kWaves][len], len = D + n_out * (D + 1)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 15, grp = lane >> 4;
  const int len = D + n_out * (D + 1);
  f32x4 wv[4], W2r[NO][4], acc_w[4], acc_W2[NO][4];
  float acc_b[NO];
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    wv[q] = *reinterpret_cast<const f32x4*>(w + 64 * q + 4 * sub);
    acc_w[q] = zero;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      W2r[o][q] = o < n_out ? *reinterpret_cast<const f32x4*>(W2 + (size_t)o * D + 64 * q + 4 * sub) : zero;
      acc_W2[o][q] = zero;
    }
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) acc_b[o] = 0.0f;
  const float inv_d = 1.0f / (float)D;
  const int64_t quads = ((int64_t)B + 3) >> 2;
  const int64_t q0 = (int64_t)blockIdx.x * kWaves + wave, qs = (int64_t)gridDim.x * kWaves;
  auto load = [&](f32x4 (&v)[4], float (&dv)[NO], int64_t quad) {
    int64_t row = 4 * quad + grp;
    if (row > B - 1) row = B - 1;
    const float* xr = x + row * D + 4 * sub;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(xr + 64 * q);
#pragma unroll
    for (int o = 0; o < NO; ++o) dv[o] = o < n_out ? dl[row * n_out + o] : 0.0f;
  };
  f32x4 cur[4], nxt[4];
  float dcur[NO], dnxt[NO];
  if (q0 < quads) load(cur, dcur, q0);
  for (int64_t quad = q0; quad < quads; quad += qs) {
    if (quad + qs < quads) load(nxt, dnxt, quad + qs);
    const int64_t row = 4 * quad + grp;
    const bool ok = row < B;                               // (rows past the batch: loaded as a copy of the last row, no contribution)
    f32x4 sv[4], gv[4];
    float sq = 0.0f, dot = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s1 = silu_(cur[q][e]);
        float g = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) g += dcur[o] * W2r[o][q][e];
        sv[q][e] = s1; gv[q][e] = g;
        sq += s1 * s1;
        dot += wv[q][e] * g * s1;
      }
    sq = row16_sum(sq);
    dot = row16_sum(dot);
    const float r = rsqrtf(sq * inv_d + eps);
    const float k3 = r * r * r * inv_d * dot;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 dx;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dx[e] = (r * wv[q][e] * gv[q][e] - sv[q][e] * k3) * silu_grad_(cur[q][e]);
        if (ok) {
          acc_w[q][e] += gv[q][e] * sv[q][e] * r;
          const float y = sv[q][e] * r * wv[q][e];
#pragma unroll
          for (int o = 0; o < NO; ++o) acc_W2[o][q][e] += dcur[o] * y;
        }
      }
      if (ok) *reinterpret_cast<f32x4*>(d_x + row * D + 64 * q + 4 * sub) = dx;
    }
    if (ok) {
#pragma unroll
      for (int o = 0; o < NO; ++o) acc_b[o] += dcur[o];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
#pragma unroll
    for (int o = 0; o < NO; ++o) dcur[o] = dnxt[o];
  }
  // the four row groups hold the same columns: (g0 + g1) + (g2 + g3)
  auto fold = [](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
  float* mine = np_red + (size_t)wave * len;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int d = 64 * q + 4 * sub + e;
      const float sw = fold(acc_w[q][e]);
      if (grp == 0) mine[d] = sw;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const float s2 = fold(acc_W2[o][q][e]);
        if (grp == 0 && o < n_out) mine[D + o * D + d] = s2;
      }
    }
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const float sb = fold(acc_b[o]);                       // (every lane of a row group counted its row once: lane 0's view)
    if (lane == 0 && o < n_out) mine[D + n_out * D + o] = sb;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < len; i += 64 * kWaves) {
    float sum = np_red[i];
#pragma unroll
    for (int w2 = 1; w2 < kWaves; ++w2) sum += np_red[(size_t)w2 * len + i];
    partial[(size_t)blockIdx.x * len + i] = sum;
  }
}

// ---- training pass: a whole sub-block forward in one launch (n = 2, D = 128) ----------------------------------------------
// gates + Linear + combine of MHCBlock._sub as three launches move 1.34 GB per 262144-row micro-batch (h is read twice, the
// branch sum and the Linear's output make a round trip each: 120 + 97 + 140 us).  Here a wave carries a 16-row tile through all
// three: the rows' branch stack is loaded once and stays in registers (mhc_gates2_kernel's layout), the Linear's weights and
// the gates' read-out weights are staged in LDS ONCE per workgroup (weight-stationary: 256 workgroups x 77 KB instead of 64 KB
// per tile through L2), the tile's 16 x 128 branch sum / output cross the wave's own LDS tile, and what the backward needs
// (pre, post, mix, the read-out sums, read, the raw Linear output z) is written on the way: 0.8 GB.  Waves never wait for
// each other after the staging barrier.
constexpr int kSubPad = 132, kSubWaves = 8;
constexpr size_t kSubLdsBytes = sizeof(float) * ((size_t)128 * kSubPad + (size_t)kSubWaves * 16 * kSubPad + 256 * 8 + 256 + 128);
struct SubFwdArgs {
  const float* h; const float* norm_w; const float* gw; const float* alpha; const float* beta; const float* lw; const float* lb;
  float* pre; float* post; float* mix; float* stats; float* read; float* z; float* h_out;
  int B, sk_it, h_rs, h_bs;                                // row / branch stride of h in floats (branch stride 0: one row repeated)
};

__global__ __launch_bounds__(64 * kSubWaves) void mhc_sub_fwd_kernel(const SubFwdArgs a) {
  constexpr int D = 128, NC = 256, G = 8;
  extern __shared__ float sub_lds[];
  float* Wl = sub_lds;                                     // [128][kSubPad]   the Linear's weight, nn.Linear layout
  float* tiles = Wl + 128 * kSubPad;                       // [kSubWaves][16][kSubPad]
  float* gwl = tiles + kSubWaves * 16 * kSubPad;           // [256][8]         the gates' read-out weight
  float* nwl = gwl + 256 * G;                              // [256]            the gates' RMS weight
  float* bl = nwl + 256;                                   // [128]            the Linear's bias
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 128 * 32; i += 64 * kSubWaves)
    *reinterpret_cast<f32x4*>(&Wl[(i >> 5) * kSubPad + 4 * (i & 31)]) = *reinterpret_cast<const f32x4*>(a.lw + (size_t)(i >> 5) * D + 4 * (i & 31));
  for (int i = threadIdx.x; i < 256 * G / 4; i += 64 * kSubWaves)
    *reinterpret_cast<f32x4*>(&gwl[4 * i]) = *reinterpret_cast<const f32x4*>(a.gw + 4 * i);
  for (int i = threadIdx.x; i < 256; i += 64 * kSubWaves) nwl[i] = a.norm_w[i];
  for (int i = threadIdx.x; i < 128; i += 64 * kSubWaves) bl[i] = a.lb[i];
  __syncthreads();
  float* tb = tiles + (size_t)wave * 16 * kSubPad;
  const int sub = lane & 15, grp = lane >> 4, r = sub, qq = grp;
  const float a0 = a.alpha[0], a1 = a.alpha[1], a2 = a.alpha[2];
  float be[G];
#pragma unroll
  for (int k = 0; k < G; ++k) be[k] = a.beta[k];
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  const int n_tiles = (a.B + 15) >> 4;
  // a tile's rows: lane (grp, sub) holds columns 64 q + 4 sub .. + 3 of rows 4 grp + it
  auto load_tile = [&](f32x4 (&dst)[4][4], int tile) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int64_t row = (int64_t)tile * 16 + 4 * grp + it;
      if (row > a.B - 1) row = a.B - 1;
      const float* hr = a.h + row * a.h_rs + 4 * sub;
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[it][q] = *reinterpret_cast<const f32x4*>(hr + (q >> 1) * a.h_bs + 64 * (q & 1));
    }
  };
  const int tile0 = blockIdx.x * kSubWaves + wave, tstride = gridDim.x * kSubWaves;
  f32x4 x[4][4], xn[4][4];
  if (tile0 < n_tiles) load_tile(x, tile0);
  for (int tile = tile0; tile < n_tiles; tile += tstride) {
    const int64_t base = (int64_t)tile * 16;
    float mine[G + 1];
#pragma unroll
    for (int k = 0; k <= G; ++k) mine[k] = 0.0f;
    {
      float Hs[4][G + 1];
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int k = 0; k <= G; ++k) Hs[it][k] = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 64 * q + 4 * sub;
        const f32x4 nw = *reinterpret_cast<const f32x4*>(&nwl[c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(&gwl[(c + e) * G]);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(&gwl[(c + e) * G + 4]);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const float xv = x[it][q][e], t = nw[e] * xv;
            Hs[it][G] += xv * xv;
#pragma unroll
            for (int k = 0; k < 4; ++k) { Hs[it][k] += t * lo[k]; Hs[it][4 + k] += t * hi[k]; }
          }
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int k = 0; k <= G; ++k) {
          const float sm = row16_sum(Hs[it][k]);
          mine[k] = sub == it ? sm : mine[k];
        }
    }
    // one lane per row (sub < 4: row 4 grp + sub): the gates; the other lanes compute on zeros
    float gt[8];                                           // pre0 pre1 post0 post1 m00 m01 m10 m11
    {
      const float r_inv = 1.0f / (sqrtf(mine[G]) / sqrtf((float)NC) + 1e-6f);
      float A[2][2], u[2] = {1.0f, 1.0f}, v[2] = {1.0f, 1.0f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        gt[i] = sigmoidf_(r_inv * mine[i] * a0 + be[i]);
        gt[2 + i] = 2.0f * sigmoidf_(r_inv * mine[2 + i] * a1 + be[2 + i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) A[i][j] = exp_(r_inv * mine[4 + 2 * i + j] * a2 + be[4 + 2 * i + j]);
      }
      for (int it = 0; it < a.sk_it; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) u[i] = rcp_(A[i][0] * v[0] + A[i][1] * v[1] + 1e-8f);
#pragma unroll
        for (int j = 0; j < 2; ++j) v[j] = rcp_(A[0][j] * u[0] + A[1][j] * u[1] + 1e-8f);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) gt[4 + 2 * i + j] = u[i] * A[i][j] * v[j];
      const int64_t my_row = base + 4 * grp + sub;
      if (sub < 4 && my_row < a.B) {
        a.pre[my_row * 2] = gt[0]; a.pre[my_row * 2 + 1] = gt[1];
        a.post[my_row * 2] = gt[2]; a.post[my_row * 2 + 1] = gt[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) a.mix[my_row * 4 + k] = gt[4 + k];
#pragma unroll
        for (int k = 0; k <= G; ++k) a.stats[my_row * (G + 1) + k] = mine[k];
      }
    }
    // row (grp, it)'s gates live in lane 16 grp + it; read = pre_0 h_0 + pre_1 h_1 -> the Linear's input tile (LDS) and HBM
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = 4 * grp + it, src = ((lane & 48) + it) << 2;
      const float p0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(gt[0])));
      const float p1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(gt[1])));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 rd;
#pragma unroll
        for (int e = 0; e < 4; ++e) rd[e] = p0 * x[it][q][e] + p1 * x[it][q + 2][e];
        *reinterpret_cast<f32x4*>(&tb[lr * kSubPad + 64 * q + 4 * sub]) = rd;
        if (base + lr < a.B) *reinterpret_cast<f32x4*>(a.read + (base + lr) * D + 64 * q + 4 * sub) = rd;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // z = read W^T + b: all eight 16-column tiles, A and B operands from LDS
    f32x4 acc[8];
    {
      f32x4 av[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const f32x4*>(&tb[r * kSubPad + 16 * j + 4 * qq]);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t] = zero;
        const float* wrow = &Wl[(16 * t + r) * kSubPad + 4 * qq];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], wv[e], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                 // one column tile's 32 weight registers at a time
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float bv = bl[16 * t + r];
#pragma unroll
      for (int g = 0; g < 4; ++g) tb[(4 * qq + g) * kSubPad + 16 * t + r] = acc[t][g] + bv;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);                     // (not earlier: with the MFMA phase's registers live it would spill)
    load_tile(xn, min(tile + tstride, n_tiles - 1));       // the next tile's rows travel during this tile's epilogue
    __builtin_amdgcn_sched_barrier(0);
    // back in the row view: z out, h'_i = post_i SiLU(z) + mix_i0 h_0 + mix_i1 h_1
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = 4 * grp + it, src = ((lane & 48) + it) << 2;
      const bool ok = base + lr < a.B;
      float gr[8];                                         // (the row's post and mix; [0], [1] unused)
#pragma unroll
      for (int k = 2; k < 8; ++k) gr[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(gt[k])));
      f32x4 o[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 zt = *reinterpret_cast<const f32x4*>(&tb[lr * kSubPad + 64 * q + 4 * sub]);
        if (ok) *reinterpret_cast<f32x4*>(a.z + (base + lr) * D + 64 * q + 4 * sub) = zt;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[q][e] = silu_(zt[e]);
      }
      if (ok) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            f32x4 hn;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              hn[e] = gr[2 + i] * o[q][e] + (gr[4 + 2 * i] * x[it][q][e] + gr[5 + 2 * i] * x[it][q + 2][e]);
            *reinterpret_cast<f32x4*>(a.h_out + ((base + lr) * 2 + i) * D + 64 * q + 4 * sub) = hn;
          }
      }
    }
    __builtin_amdgcn_wave_barrier();                       // the next tile's branch sums overwrite this wave's LDS tile
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int q = 0; q < 4; ++q) x[it][q] = xn[it][q];
  }
}

// ---- training pass: a whole sub-block BACKWARD in one launch (n = 2, D = 128) ---------------------------------------------
// As four launches (combine_bwd, the Linear's input gradient, read_bwd, gates_bwd) the backward of one sub-block streams the
// upstream gradient g and the branch stack h through HBM three times each and makes round trips of d_z, d_read, d_pre, d_post,
// d_mix: 2.5 GB and 536 us per 262144-row micro-batch.  Here a wave carries a 16-row tile through all four (the forward
// kernel's layout: lane (grp, sub) holds columns 64 q + 4 sub .. + 3 of rows 4 grp + it, so both branches of a column live
// in one lane and a row's sums are four DPP adds):
//   P1  d_z = SiLU'(z) sum_i post_i g_i -> the wave's LDS tile (+ HBM, for the Linear's weight gradient), and the row sums
//       d_post_i = <g_i, SiLU(z)>, d_mix_ij = <g_i, h_j>;
//   P2  d_read = d_z W on f32 MFMA (A from the tile, B from the transposed weight staged in LDS once per workgroup), back
//       through the tile into the row view; d_pre_i = <d_read, h_i>;
//   P3  the gates' backward for the row (one lane per row: sigmoid' / exp' from the saved outputs, the RMS statistic's path),
//       broadcast to the row's lanes;
//   P4  d_h = norm_w (dH w^T) + d|flat| flat / |flat| + pre_j d_read + sum_i mix_ij g_i, written once (optionally summed over
//       the branches: the first sub-block's input is one row repeated), and the parameter sums: d norm_w per lane, d w as a
//       [256 x rows] x [rows x 8] product on MFMA (A = norm_w * h from the registers, B = dH selected by lane), d alpha / d beta
//       in the row lanes.  One partial vector per workgroup, added ascending by partial_reduce_kernel: no atomics.
// g and h are read once (1 KB each per row), z once, d_z and d_h written once: 1.1 GB.  `g` / `h` may be broadcast over the
// branches (branch stride 0): the last sub-block's upstream gradient is the final norm's d x for both branches.
constexpr int kSubBwdWaves = 4;
#ifdef GYMRL_PROF_BUILD
// probe build only: shader-clock cycles per phase, summed over wave 0 of every workgroup (tools/micro_sub_bwd.py --phases)
__device__ unsigned long long g_sub_bwd_prof[8];
#define SUB_MARK(k) do { const long long now_ = (long long)__builtin_readcyclecounter(); prof_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define SUB_MARK(k) do {} while (0)
#endif
constexpr size_t kSubBwdLdsBytes = sizeof(float) * ((size_t)128 * kSubPad + (size_t)kSubBwdWaves * 16 * kSubPad + 256);
static_assert((size_t)kSubBwdWaves * kGatesLen <= (size_t)128 * kSubPad, "the partial sums reuse the weight's LDS");
struct SubBwdArgs {
  const float* g; const float* h; const float* z;
  const float* pre; const float* post; const float* mix; const float* stats;
  const float* norm_w; const float* gw; const float* alpha; const float* lw;
  float* d_z; float* d_h; float* partial;
  int B, g_rs, g_bs, h_rs, h_bs;                           // row / branch strides of g and h in floats
};

__device__ __forceinline__ float lane_bcast(int src_byte, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_byte, __float_as_int(v)));
}

template <bool SUM_DH>
__global__ __launch_bounds__(64 * kSubBwdWaves) void mhc_sub_bwd_kernel(const SubBwdArgs a) {
  constexpr int D = 128, G = 8;
  extern __shared__ float sub_lds[];
  float* Wt = sub_lds;                                     // [128][kSubPad]   Wt[k][n] = W[n][k]
  float* tiles = Wt + 128 * kSubPad;                       // [kSubBwdWaves][16][kSubPad]
  float* nwl = tiles + kSubBwdWaves * 16 * kSubPad;        // [256]            the gates' RMS weight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 128 * 32; i += 64 * kSubBwdWaves) {
    const int n = i & 127, c = i >> 7;                     // lanes walk n: conflict-free LDS writes (the reads are 64 KB from L2, once)
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.lw + (size_t)n * D + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) Wt[(4 * c + e) * kSubPad + n] = v[e];
  }
  for (int i = threadIdx.x; i < 256; i += 64 * kSubBwdWaves) nwl[i] = a.norm_w[i];
  __syncthreads();
  float* tb = tiles + (size_t)wave * 16 * kSubPad;
  const int sub = lane & 15, grp = lane >> 4;
  // the gates' read-out weight of this lane's 16 columns, for the whole launch, as the B operand of t2 = dH w^T (two k-steps
  // of four gates: lane (grp, sub) holds w[column(q, e, sub)][4 s + grp])
  float wB[4][4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wB[q][e][ks] = a.gw[(size_t)(64 * q + 4 * sub + e) * G + 4 * ks + grp];
  }
  const float al[3] = {a.alpha[0], a.alpha[1], a.alpha[2]};
  const float inv_sqrt_nc = 1.0f / sqrtf(256.0f);
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc_w[4][4];                                       // [q][e]: d w[64 q + 4 (4 (lane >> 4) + g) + e][lane & 15]
  float acc_nw[4][4], acc_al[3] = {0.0f, 0.0f, 0.0f}, acc_be[G];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc_w[q][e] = zero; acc_nw[q][e] = 0.0f; }
#pragma unroll
  for (int k = 0; k < G; ++k) acc_be[k] = 0.0f;
  const int n_tiles = (a.B + 15) >> 4;
#ifdef GYMRL_PROF_BUILD
  long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = (long long)__builtin_readcyclecounter();
#endif
  // The tile loop is a software pipeline in the wave's own registers (one wave per SIMD: nobody else hides its latencies):
  //   g, z and the row lanes' scalars of tile i + 1 are requested inside P4 of tile i, each half as soon as P4 has used
  //   the registers it lands in; h of tile i is requested before the MFMA phase and first used after it (the d_mix sums
  //   moved there from P1 for that).  Loads in flight: 24 KB under P4's second half, 16 KB under the 256 MFMAs.
  const int tstride = gridDim.x * kSubBwdWaves;
  f32x4 gv[4][4], zv[4][2];
  f32x2 pre2, post2;
  f32x4 mix4;
  float st[G + 1];
  // rows past the end read the last row instead and are switched off in P3 (their dH = 0: nothing of them reaches a sum or a store)
  auto load_gz = [&](int t, int q2) {                      // columns 64 q2 .. 64 q2 + 63 of g (both branches) and z
    if (t > n_tiles - 1) t = n_tiles - 1;
    const int64_t nbase = (int64_t)t * 16;
    const int nlast = (int)(a.B - 1 - nbase);
    const float* gt = a.g + nbase * a.g_rs;                // (uniform: the tile's base; the lane's offsets are 32-bit)
    const float* zt = a.z + nbase * D;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = 4 * grp + it, rc = lr <= nlast ? lr : nlast;
      const int off = 64 * q2 + 4 * sub;
      gv[it][q2] = *reinterpret_cast<const f32x4*>(gt + (rc * a.g_rs + off));
      gv[it][q2 + 2] = *reinterpret_cast<const f32x4*>(gt + (rc * a.g_rs + a.g_bs + off));
      zv[it][q2] = *reinterpret_cast<const f32x4*>(zt + (rc * D + off));
    }
  };
  auto load_scalars = [&](int t) {                         // lane 16 grp + it owns row 4 grp + it
    if (t > n_tiles - 1) t = n_tiles - 1;
    const int64_t nbase = (int64_t)t * 16;
    const int nlast = (int)(a.B - 1 - nbase), own_l = 4 * grp + (sub & 3);
    const int64_t own_c = nbase + (own_l <= nlast ? own_l : nlast);
    pre2 = *reinterpret_cast<const f32x2*>(a.pre + own_c * 2);
    post2 = *reinterpret_cast<const f32x2*>(a.post + own_c * 2);
    mix4 = *reinterpret_cast<const f32x4*>(a.mix + own_c * 4);
#pragma unroll
    for (int k = 0; k <= G; ++k) st[k] = a.stats[own_c * (G + 1) + k];
  };
  const int tile0 = blockIdx.x * kSubBwdWaves + wave;
  if (tile0 < n_tiles) { load_scalars(tile0); load_gz(tile0, 0); load_gz(tile0, 1); }
  for (int tile = tile0; tile < n_tiles; tile += tstride) {
    const int64_t base = (int64_t)tile * 16;
    SUB_MARK(7);
    const int last = (int)(a.B - 1 - base);                // >= 0: the tile's last valid local row (or beyond 15)
    const bool own_in = 4 * grp + (sub & 3) <= last, own_ok = sub < 4 && own_in;
    // ---- P1: d_z and d_post (what needs g and z only); columns 0..63 of every row first: their loads were requested first
    float dsum[4][6];                                      // per row: d_post_0, d_post_1, d_mix_00, _01, _10, _11
    {
      float po[4][2], s0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int src = ((lane & 48) + it) << 2;
        po[it][0] = lane_bcast(src, post2[0]); po[it][1] = lane_bcast(src, post2[1]);
      }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int lr = 4 * grp + it;
          f32x4 dzv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zt = zv[it][q2][e], sg = sigmoidf_(zt), o = zt * sg;
            const float g0 = gv[it][q2][e], g1 = gv[it][q2 + 2][e];
            dzv[e] = (po[it][0] * g0 + po[it][1] * g1) * (sg * (1.0f + zt * (1.0f - sg)));
            s0[it] += g0 * o; s1[it] += g1 * o;
          }
          *reinterpret_cast<f32x4*>(&tb[lr * kSubPad + 64 * q2 + 4 * sub]) = dzv;
          *reinterpret_cast<f32x4*>(a.d_z + (base + lr) * D + 64 * q2 + 4 * sub) = dzv;   // (the outputs are padded to whole tiles)
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) { dsum[it][0] = row16_sum(s0[it]); dsum[it][1] = row16_sum(s1[it]); }
    }
    SUB_MARK(1);                                           // P1, incl. the wait for g and z
    // the branch stack: requested now, first used after the MFMA phase
    f32x4 hv[4][4];
    {
      const float* ht = a.h + base * a.h_rs;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int lr = 4 * grp + it, rc = lr <= last ? lr : last;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          hv[it][q] = *reinterpret_cast<const f32x4*>(ht + (rc * a.h_rs + (q >> 1) * a.h_bs + 64 * (q & 1) + 4 * sub));
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- P2: d_read = d_z W, all eight 16-column tiles (the forward's loop with the transposed weight)
    f32x4 dr[4][2];
    {
      f32x4 acc[8], av[8], wv[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const f32x4*>(&tb[sub * kSubPad + 16 * j + 4 * grp]);
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[0][j] = *reinterpret_cast<const f32x4*>(&Wt[sub * kSubPad + 4 * grp + 16 * j]);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t] = zero;
        if (t < 7) {                                       // the next column tile's weights travel under this tile's 32 MFMAs
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[(t + 1) & 1][j] = *reinterpret_cast<const f32x4*>(&Wt[(16 * (t + 1) + sub) * kSubPad + 4 * grp + 16 * j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], wv[t & 1][j][e], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // two column tiles' weight registers at a time, not all eight
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) tb[(4 * grp + gq) * kSubPad + 16 * t + sub] = acc[t][gq];
    }
    __builtin_amdgcn_wave_barrier();
    SUB_MARK(2);                                           // P2: the MFMA phase and the tile's way back into LDS
    // d_read back in the row view; the sums that need h: d_pre_i = <d_read, h_i>, d_mix_ij = <g_i, h_j>
    float dpre[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float s[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        dr[it][q2] = *reinterpret_cast<const f32x4*>(&tb[(4 * grp + it) * kSubPad + 64 * q2 + 4 * sub]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g0 = gv[it][q2][e], g1 = gv[it][q2 + 2][e], h0 = hv[it][q2][e], h1 = hv[it][q2 + 2][e];
          s[0] += dr[it][q2][e] * h0; s[1] += dr[it][q2][e] * h1;
          s[2] += g0 * h0; s[3] += g0 * h1; s[4] += g1 * h0; s[5] += g1 * h1;
        }
      }
      dpre[it][0] = row16_sum(s[0]); dpre[it][1] = row16_sum(s[1]);
#pragma unroll
      for (int v = 0; v < 4; ++v) dsum[it][2 + v] = row16_sum(s[2 + v]);
    }
    __builtin_amdgcn_wave_barrier();                       // the rows' scalars overwrite d_read in this wave's LDS tile
    SUB_MARK(3);                                           // d_read back in the row view, the sums with h (incl. the wait for h)
    // ---- P3: the row lane's gates backward (mhc_gates_bwd_kernel's phase A)
    float dH[G], dn_over;
    {
      float up[G];                                         // d_pre 0 1, d_post 0 1, d_mix 00 01 10 11 of this lane's row
#pragma unroll
      for (int k = 0; k < G; ++k) up[k] = 0.0f;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const bool me = (sub & 3) == it;
        up[0] = me ? dpre[it][0] : up[0]; up[1] = me ? dpre[it][1] : up[1];
#pragma unroll
        for (int v = 0; v < 6; ++v) up[2 + v] = me ? dsum[it][v] : up[2 + v];
      }
      const float norm = sqrtf(st[G]);
      const float r = 1.0f / (norm * inv_sqrt_nc + 1e-6f);
      float dz[G];
      dz[0] = up[0] * pre2[0] * (1.0f - pre2[0]);
      dz[1] = up[1] * pre2[1] * (1.0f - pre2[1]);
      dz[2] = up[2] * post2[0] * (1.0f - 0.5f * post2[0]);
      dz[3] = up[3] * post2[1] * (1.0f - 0.5f * post2[1]);
#pragma unroll
      for (int k = 0; k < 4; ++k) dz[4 + k] = up[4 + k] * mix4[k];
      if (!own_in) {
#pragma unroll
        for (int k = 0; k < G; ++k) dz[k] = 0.0f;
      }
      float d_r = 0.0f;
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int gi = k < 2 ? 0 : (k < 4 ? 1 : 2);
        dH[k] = dz[k] * r * al[gi];
        d_r += dz[k] * st[k] * al[gi];
        if (own_ok) { acc_al[gi] += dz[k] * r * st[k]; acc_be[k] += dz[k]; }
      }
      const float d_norm = d_r * (-r * r * inv_sqrt_nc);
      dn_over = norm > 0.0f ? d_norm / norm : 0.0f;
    }
    SUB_MARK(4);                                           // P3
    // the row lanes publish their row's sixteen scalars through the wave's LDS tile (free again: d_read is in registers)
    if (sub < 4) {
      float* rowp = &tb[(4 * grp + sub) * kSubPad];
      *reinterpret_cast<f32x4*>(rowp) = f32x4{dH[0], dH[1], dH[2], dH[3]};
      *reinterpret_cast<f32x4*>(rowp + 4) = f32x4{dH[4], dH[5], dH[6], dH[7]};
      *reinterpret_cast<f32x4*>(rowp + 8) = f32x4{dn_over, pre2[0], pre2[1], 0.0f};
      *reinterpret_cast<f32x4*>(rowp + 12) = mix4;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- P4: d_h and the column sums.  t2[row][column] = sum_k dH[row][k] w[column][k] is a [16 x 8] x [8 x 256] product:
    // two MFMAs per 16 columns (A = dH[row = lane & 15][4 s + (lane >> 4)], B = the lane's weight registers); element `it` of
    // the result is row 4 grp + it of this lane's column — the row view's own layout
    const float aH0 = tb[sub * kSubPad + grp], aH1 = tb[sub * kSubPad + 4 + grp];
    float sdn[4], sp[4][2], sm[4][4], bsel[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float* rowp = &tb[(4 * grp + it) * kSubPad];
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(rowp + 8), c1 = *reinterpret_cast<const f32x4*>(rowp + 12);
      sdn[it] = c0[0]; sp[it][0] = c0[1]; sp[it][1] = c0[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) sm[it][k] = c1[k];
      const float b = rowp[sub & 7];                       // the d w MFMA's B operand: dH[row][gate = lane & 15], 0 beyond the 8 gates
      bsel[it] = sub < G ? b : 0.0f;
    }
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      f32x4 dx0[4];                                        // (sum_dh) branch 0's gradient of these columns
#pragma unroll
      for (int br = 0; br < 2; ++br) {
        const int q = 2 * br + q2;
        f32x4 dx[4];
        const f32x4 nw = *reinterpret_cast<const f32x4*>(&nwl[64 * q + 4 * sub]);
        f32x4 t2v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          t2v[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aH0, wB[q][e][0], zero, 0, 0, 0);
          t2v[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(aH1, wB[q][e][1], t2v[e], 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {                    // (four accumulators in turn: no MFMA waits for the one before it)
            const float x = hv[it][q][e], t2 = t2v[e][it];
            dx[it][e] = nw[e] * t2 + sdn[it] * x + sp[it][br] * dr[it][q2][e] +
                        (sm[it][br] * gv[it][q2][e] + sm[it][2 + br] * gv[it][q2 + 2][e]);
            acc_nw[q][e] += x * t2;
            acc_w[q][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(nw[e] * x, bsel[it], acc_w[q][e], 0, 0, 0);
          }
          // rows past the end of the batch are written too: d_z and d_h are padded to whole 16-row tiles
          float* dst = SUM_DH ? a.d_h + (base + 4 * grp + it) * D + 64 * q2 + 4 * sub
                              : a.d_h + ((base + 4 * grp + it) * 2 + br) * D + 64 * q2 + 4 * sub;
          if constexpr (!SUM_DH) *reinterpret_cast<f32x4*>(dst) = dx[it];
          else if (br == 0) dx0[it] = dx[it];
          else *reinterpret_cast<f32x4*>(dst) = dx0[it] + dx[it];   // d of a repeated row: the branches' gradients added
        }
      }
      __builtin_amdgcn_sched_barrier(0);                   // (the loads below must not move up over the last use of their registers)
      load_gz(tile + tstride, q2);                         // these columns of g and z are done: the next tile's are requested
      if (q2 == 0) load_scalars(tile + tstride);           // (P1 starts with them: not at the very end)
      __builtin_amdgcn_sched_barrier(0);
    }
    SUB_MARK(5);                                           // P4
    __builtin_amdgcn_wave_barrier();                       // the next tile's d_z overwrites the rows' scalars
    SUB_MARK(6);
  }
#ifdef GYMRL_PROF_BUILD
  if (threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_sub_bwd_prof[k], (unsigned long long)prof_[k]);
#endif
  // ---- the workgroup's partial: rows across the wave, then the four waves in a fixed order
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc_nw[q][e] += __shfl_xor(acc_nw[q][e], 16, 64);
      acc_nw[q][e] += __shfl_xor(acc_nw[q][e], 32, 64);
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) acc_al[gi] += __shfl_xor(acc_al[gi], off, 64);
#pragma unroll
    for (int k = 0; k < G; ++k) acc_be[k] += __shfl_xor(acc_be[k], off, 64);
  }
  __syncthreads();                                         // every wave is done with the weight: its LDS holds the partials now
  float* mine = sub_lds + (size_t)wave * kGatesLen;
  if (grp == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) mine[64 * q + 4 * sub + e] = acc_nw[q][e];
  }
  if (sub < G) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) mine[256 + (64 * q + 4 * (4 * grp + gq) + e) * G + sub] = acc_w[q][e][gq];
  }
  if (lane == 0) {
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) mine[256 + 256 * G + gi] = acc_al[gi];
#pragma unroll
    for (int k = 0; k < G; ++k) mine[256 + 256 * G + 3 + k] = acc_be[k];
  }
  __syncthreads();
  float* out = a.partial + (size_t)blockIdx.x * kGatesLen;
  for (int i = threadIdx.x; i < kGatesLen; i += 64 * kSubBwdWaves) {
    float sum = sub_lds[i];
#pragma unroll
    for (int w2 = 1; w2 < kSubBwdWaves; ++w2) sum += sub_lds[(size_t)w2 * kGatesLen + i];
    out[i] = sum;
  }
}

// ---- the whole rollout forward of PPO-full's network in ONE launch: mhc_policy_device.hpp's 16-row tile per workgroup --------
// gymrl_mhc_policy_pack: one thread per float of the image (mhc_policy_device.hpp: layout next to PolicyArgs)
__global__ __launch_bounds__(256) void mhc_policy_pack_kernel(const PolicyArgs a, float* __restrict__ img) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t subs = (size_t)a.n_sub * kPolSubFloats;
  if (idx >= subs + 2 * (size_t)kPolHeadFloats) return;
  if (idx < subs) {
    const int s = (int)(idx / kPolSubFloats);
    int o = (int)(idx - (size_t)s * kPolSubFloats);
    if (o < kPolLwFloats) {                                  // ((T * 8 + j) * 64 + lane) * 4 + c
      const int c = o & 3, lane = (o >> 2) & 63, j = (o >> 8) & 7, T = o >> 11;
      img[idx] = a.lw[s][(size_t)(16 * T + (lane & 15)) * 128 + 16 * j + 4 * (lane >> 4) + c];
    } else {                                                 // (((q * 4 + e) * 2 + half) * 16 + sub) * 4 + k
      o -= kPolLwFloats;
      const int k = o & 3, sub = (o >> 2) & 15, half = (o >> 6) & 1, e = (o >> 7) & 3, q = o >> 9;
      img[idx] = a.gw[s][(size_t)(64 * q + 4 * sub + e) * 8 + 4 * half + k];
    }
    return;
  }
  int o = (int)(idx - subs);
  const int hd = o / kPolHeadFloats;
  o -= hd * kPolHeadFloats;
  const int c = o & 3, lane = (o >> 2) & 63, j = (o >> 8) & 7, T = o >> 11;
  img[idx] = a.h1_w[hd][(size_t)(16 * T + (lane & 15)) * 128 + 16 * j + 4 * (lane >> 4) + c];
}

__global__ __launch_bounds__(256) void mhc_policy_kernel(const PolicyArgs a, const float* __restrict__ obs, int B,
                                                         float* __restrict__ logits, float* __restrict__ value) {
  __shared__ PolicyLds L;
  int64_t row = (int64_t)blockIdx.x * 16 + 4 * (threadIdx.x >> 6) + ((threadIdx.x & 63) >> 4);
  if (row > B - 1) row = B - 1;
  const int left = B - (int)blockIdx.x * 16;
  policy_tile(a, L, obs + row * a.obs_dim, logits + (size_t)blockIdx.x * 16 * a.n_act, a.n_act, value + (size_t)blockIdx.x * 16, 1,
              left < 16 ? left : 16);
}

}  // namespace

extern "C" {

int gymrl_mhc_gates(const float* h, const float* norm_w, const float* w, const float* alpha, const float* beta, int B, int n,
                    int D, int sk_it, float* pre_out, float* post_out, float* mix_out, float* read_out, float* stats_out,
                    void* stream) {
  if (!h || !norm_w || !w || !alpha || !beta || !pre_out || !post_out || !mix_out || !read_out || B < 0 || D < 4 || D % 4 ||
      sk_it < 0 || (n != 2 && n != 4))
    return -22;
  const bool batched = n == 2 && (n * D == 256 || n * D == 512);
  if (stats_out && !batched) return -22;
  if (B == 0) return 0;
  GatesArgs a{h, norm_w, w, alpha, beta, pre_out, post_out, mix_out, read_out, stats_out, B, D, sk_it};
  if (batched) {
    const int rb = n * D == 256 ? 16 : 8;                  // rows per wave
    const dim3 grid((B + rb - 1) / rb), block(64);
    if (n * D == 256) hipLaunchKernelGGL(mhc_gates2_kernel<1>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mhc_gates2_kernel<2>, grid, block, 0, (hipStream_t)stream, a);
  } else {
    const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
    if (n == 2) hipLaunchKernelGGL(mhc_gates_kernel<2>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mhc_gates_kernel<4>, grid, block, 0, (hipStream_t)stream, a);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_combine(const float* post, const float* mix, const float* out, const float* h, int B, int n, int D, int act,
                      float* h_out, void* stream) {
  if (!post || !mix || !out || !h || !h_out || B < 0 || D < 1 || (n != 2 && n != 4) || (act != GYMRL_ACT_NONE && act != GYMRL_ACT_SILU))
    return -22;
  if (B == 0) return 0;
  int64_t nb = ((int64_t)B * D + 255) / 256;
  if (nb > 4096) nb = 4096;
  const int silu = act == GYMRL_ACT_SILU;
  if (n == 2) hipLaunchKernelGGL(mhc_combine_kernel<2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, post, mix, out, h, B, D, silu, h_out);
  else hipLaunchKernelGGL(mhc_combine_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, post, mix, out, h, B, D, silu, h_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_read_fwd(const float* pre, const float* h, int B, int n, int D, float* read_out, void* stream) {
  if (!pre || !h || !read_out || B < 0 || D < 4 || D % 4 || (n != 2 && n != 4)) return -22;
  if (B == 0) return 0;
  int64_t nb = ((int64_t)B * (D / 4) + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (n == 2) hipLaunchKernelGGL(mhc_read_fwd_kernel<2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pre, h, B, D, read_out);
  else hipLaunchKernelGGL(mhc_read_fwd_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pre, h, B, D, read_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_read_bwd(const float* g, const float* pre, const float* h, int B, int n, int D, float* d_pre, float* d_h,
                       int accumulate, void* stream) {
  if (!g || !pre || !h || !d_pre || B < 0 || D < 1 || (n != 2 && n != 4)) return -22;
  if (B == 0) return 0;
  const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
  if (n == 2) hipLaunchKernelGGL(mhc_read_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g, pre, h, B, D, d_pre, d_h, accumulate);
  else hipLaunchKernelGGL(mhc_read_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, g, pre, h, B, D, d_pre, d_h, accumulate);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_combine_bwd(const float* g, const float* post, const float* mix, const float* out, const float* h, int B, int n, int D,
                          int act, float* d_post, float* d_mix, float* d_out, float* d_h, void* stream) {
  if (!g || !post || !mix || !out || !h || !d_post || !d_mix || !d_out || B < 0 || D < 1 || (n != 2 && n != 4) ||
      (act != GYMRL_ACT_NONE && act != GYMRL_ACT_SILU))
    return -22;
  if (B == 0) return 0;
  const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
  const int silu = act == GYMRL_ACT_SILU;
  if (n == 2) hipLaunchKernelGGL(mhc_combine_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g, post, mix, out, h, B, D, silu, d_post, d_mix, d_out, d_h);
  else hipLaunchKernelGGL(mhc_combine_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, g, post, mix, out, h, B, D, silu, d_post, d_mix, d_out, d_h);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

static int gates_bwd_blocks(int B) {
  int blocks = (B + 64 * kWaves - 1) / (64 * kWaves);     // 512 = every wave slot of the chip at the kernel's 2 waves per SIMD
  return blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
}

size_t gymrl_mhc_gates_bwd_workspace_bytes(int n, int D) {
  const int ch = n * D / 256;
  return sizeof(float) * 1024 * (size_t)(ch < 1 ? 1 : ch) * kGatesLen;
}

int gymrl_mhc_gates_bwd(const float* h, const float* norm_w, const float* w, const float* alpha, const float* pre, const float* post,
                        const float* mix, const float* stats, const float* d_pre, const float* d_post, const float* d_mix,
                        const float* d_read, const float* g_out, int B, int n, int D, float* d_h, float* d_norm_w, float* d_w,
                        float* d_alpha, float* d_beta, void* workspace, void* stream) {
  if (!h || !norm_w || !w || !alpha || !pre || !post || !mix || !stats || !d_pre || !d_post || !d_mix || !d_h || !d_norm_w || !d_w ||
      !d_alpha || !d_beta || !workspace || B < 1 || n != 2 || (n * D != 256 && n * D != 512))
    return -22;
  const int ch = n * D / 256, blocks = gates_bwd_blocks(B);
  GatesBwdArgs a{h, norm_w, w, alpha, pre, post, mix, stats, d_pre, d_post, d_mix, d_read, g_out, d_h, static_cast<float*>(workspace),
                 B, D};
  hipLaunchKernelGGL(mhc_gates_bwd_kernel, dim3(blocks, ch), dim3(64 * kWaves), sizeof(float) * kWaves * kGatesLen,
                     (hipStream_t)stream, a);
  ReduceArgs r{static_cast<const float*>(workspace), blocks, kGatesLen, 4,
               {256, 256 + 256 * 8, 256 + 256 * 8 + 3, kGatesLen}, {256, 256 * 8, 0, 0}, {d_norm_w, d_w, d_alpha, d_beta}};
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((kGatesLen + 31) / 32, ch), dim3(256), 0, (hipStream_t)stream, r);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sinkhorn(const float* A, int B, int n, int sk_it, float* u_out, float* v_out, void* stream) {
  if (!A || !u_out || !v_out || B < 0 || sk_it < 0 || (n != 2 && n != 4)) return -22;
  if (B == 0) return 0;
  const dim3 grid((B + 255) / 256), block(256);
  if (n == 2) hipLaunchKernelGGL(sinkhorn_kernel<2>, grid, block, 0, (hipStream_t)stream, A, B, sk_it, u_out, v_out);
  else hipLaunchKernelGGL(sinkhorn_kernel<4>, grid, block, 0, (hipStream_t)stream, A, B, sk_it, u_out, v_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_rmsnorm(const float* x, const float* w, int B, int D, int n_sum, float eps, int act, float* y, void* stream) {
  if (!x || !w || !y || B < 0 || D < 1 || n_sum < 1 || (act != GYMRL_ACT_NONE && act != GYMRL_ACT_SILU)) return -22;
  if (B == 0) return 0;
  const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
  const dim3 walk(grid.x > 4096 ? 4096 : grid.x);          // the register-resident kernels: every wave slot of the chip, rows in a loop
  const int silu = act == GYMRL_ACT_SILU;
  hipStream_t s = (hipStream_t)stream;
  if (D <= 128) hipLaunchKernelGGL(rmsnorm_reg_kernel<2>, walk, block, 0, s, x, w, B, D, n_sum, eps, silu, y);
  else if (D <= 256) hipLaunchKernelGGL(rmsnorm_reg_kernel<4>, walk, block, 0, s, x, w, B, D, n_sum, eps, silu, y);
  else if (D <= 512) hipLaunchKernelGGL(rmsnorm_reg_kernel<8>, walk, block, 0, s, x, w, B, D, n_sum, eps, silu, y);
  else hipLaunchKernelGGL(rmsnorm_kernel, grid, block, 0, s, x, w, B, D, n_sum, eps, silu, y);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

static int rmsnorm_bwd_blocks(int B) {
  int blocks = (B + kWaves - 1) / kWaves;
  return blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
}

size_t gymrl_rmsnorm_bwd_workspace_bytes(int D) { return sizeof(float) * 2048 * (size_t)(D < 1 ? 1 : D); }

int gymrl_rmsnorm_sum_bwd(const float* g, const float* x, const float* w, int B, int D, int n_sum, float eps, int act, float* d_x,
                          float* d_w, void* workspace, void* stream);

int gymrl_rmsnorm_bwd(const float* g, const float* x, const float* w, int B, int D, float eps, int act, float* d_x, float* d_w,
                      void* workspace, void* stream) {
  return gymrl_rmsnorm_sum_bwd(g, x, w, B, D, 1, eps, act, d_x, d_w, workspace, stream);
}

int gymrl_rmsnorm_sum_bwd(const float* g, const float* x, const float* w, int B, int D, int n_sum, float eps, int act, float* d_x,
                          float* d_w, void* workspace, void* stream) {
  if (!g || !x || !w || !d_x || !d_w || !workspace || B < 1 || D < 1 || D > 512 || n_sum < 1 ||
      (act != GYMRL_ACT_NONE && act != GYMRL_ACT_SILU))
    return -22;
  const int blocks = rmsnorm_bwd_blocks(B), silu = act == GYMRL_ACT_SILU;
  float* part = static_cast<float*>(workspace);
  const dim3 grid(blocks), block(64 * kWaves);
  if (D <= 128) hipLaunchKernelGGL(rmsnorm_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g, x, w, B, D, n_sum, eps, silu, d_x, part);
  else if (D <= 256) hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, g, x, w, B, D, n_sum, eps, silu, d_x, part);
  else hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, grid, block, 0, (hipStream_t)stream, g, x, w, B, D, n_sum, eps, silu, d_x, part);
  ReduceArgs r{part, blocks, D, 1, {D, 0, 0, 0}, {0, 0, 0, 0}, {d_w, nullptr, nullptr, nullptr}};
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((D + 31) / 32, 1), dim3(256), 0, (hipStream_t)stream, r);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

#define NORM_PROJ_DISPATCH(KERNEL, ...)                                                                            \
  do {                                                                                                             \
    if (D <= 128) {                                                                                                \
      if (n_out <= 1) hipLaunchKernelGGL((KERNEL<2, 1>), __VA_ARGS__);                                            \
      else if (n_out <= 4) hipLaunchKernelGGL((KERNEL<2, 4>), __VA_ARGS__);                                       \
      else hipLaunchKernelGGL((KERNEL<2, 8>), __VA_ARGS__);                                                        \
    } else {                                                                                                       \
      if (n_out <= 1) hipLaunchKernelGGL((KERNEL<4, 1>), __VA_ARGS__);                                            \
      else if (n_out <= 4) hipLaunchKernelGGL((KERNEL<4, 4>), __VA_ARGS__);                                       \
      else hipLaunchKernelGGL((KERNEL<4, 8>), __VA_ARGS__);                                                        \
    }                                                                                                              \
  } while (0)

int gymrl_norm_proj_fwd(const float* x, const float* norm_w, const float* W2, const float* b2, int B, int D, int n_out, float eps,
                        float* out, void* stream) {
  if (!x || !norm_w || !W2 || !out || B < 0 || D < 1 || D > 256 || n_out < 1 || n_out > 8) return -22;
  if (B == 0) return 0;
  const dim3 block(64 * kWaves);
  const unsigned want = (unsigned)((B + kWaves - 1) / kWaves);
  const dim3 grid(want > 4096 ? 4096 : want);
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // The kernel — and with it the summation order, i.e. the result's last bits — is chosen by SHAPE alone: D = 256 takes the
  // four-row kernel and therefore REQUIRES 16-byte aligned operands (-22 otherwise: the caller copies to an aligned buffer);
  // a choice by pointer alignment would make the bits depend on where an allocation or a view happens to start.
  if (D == 256 && !(al16(x) && al16(norm_w) && al16(W2))) return -22;
  if (D == 256) {                                                   // four rows per wave (16-byte loads, 16-lane sums)
    const unsigned wq = (unsigned)(((B + 3) / 4 + kWaves - 1) / kWaves);
    const dim3 g4(wq > 2048 ? 2048 : wq);
    if (n_out <= 1) hipLaunchKernelGGL(norm_proj_fwd4_kernel<1>, g4, block, 0, (hipStream_t)stream, x, norm_w, W2, b2, B, n_out, eps, out);
    else if (n_out <= 4) hipLaunchKernelGGL(norm_proj_fwd4_kernel<4>, g4, block, 0, (hipStream_t)stream, x, norm_w, W2, b2, B, n_out, eps, out);
    else hipLaunchKernelGGL(norm_proj_fwd4_kernel<8>, g4, block, 0, (hipStream_t)stream, x, norm_w, W2, b2, B, n_out, eps, out);
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  NORM_PROJ_DISPATCH(norm_proj_fwd_kernel, grid, block, 0, (hipStream_t)stream, x, norm_w, W2, b2, B, D, n_out, eps, out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

size_t gymrl_norm_proj_bwd_workspace_bytes(int D, int n_out) {
  return sizeof(float) * 2048 * ((size_t)(D < 1 ? 1 : D) * (size_t)((n_out < 1 ? 1 : n_out) + 1) + (size_t)(n_out < 1 ? 1 : n_out));
}

int gymrl_norm_proj_bwd(const float* d_out, const float* x, const float* norm_w, const float* W2, int B, int D, int n_out, float eps,
                        float* d_x, float* d_norm_w, float* d_W2, float* d_b2, void* workspace, void* stream) {
  if (!d_out || !x || !norm_w || !W2 || !d_x || !d_norm_w || !d_W2 || !d_b2 || !workspace || B < 1 || D < 1 || D > 256 || n_out < 1 ||
      n_out > 8)
    return -22;
  const int blocks = rmsnorm_bwd_blocks(B), len = D + n_out * (D + 1);
  float* part = static_cast<float*>(workspace);
  const dim3 grid(blocks), block(64 * kWaves);
  const size_t lds = sizeof(float) * (size_t)kWaves * len;
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // four rows per wave for ONE output (the critic's head: 244 -> 189 us at 524 288 rows).  With four outputs the per-lane weight and
  // accumulator vectors take 292 registers — one wave per SIMD: 410 us against the one-row kernel's 268 — so n_out > 1 stays there.
  // Chosen by shape alone (gymrl_norm_proj_fwd's rule): D = 256 with one output requires aligned operands.  The backward
  // recomputes the row's 1 / rms in ITS kernel's summation order — for n_out > 1 at D = 256 not the forward's (four-row) order:
  // the two values of r can differ in the last bit, a relative 1e-7 on the gradient, the same for every run.
  if (D == 256 && n_out == 1 && !(al16(x) && al16(norm_w) && al16(W2) && al16(d_x))) return -22;
  if (D == 256 && n_out == 1)
    hipLaunchKernelGGL(norm_proj_bwd4_kernel<1>, grid, block, lds, (hipStream_t)stream, d_out, x, norm_w, W2, B, n_out, eps, d_x, part);
  else
  NORM_PROJ_DISPATCH(norm_proj_bwd_kernel, grid, block, lds, (hipStream_t)stream, d_out, x, norm_w, W2, B, D, n_out, eps, d_x, part);
  ReduceArgs r{part, blocks, len, 3, {D, D + n_out * D, len, 0}, {0, 0, 0, 0}, {d_norm_w, d_W2, d_b2, nullptr}};
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((len + 31) / 32, 1), dim3(256), 0, (hipStream_t)stream, r);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_sub_forward(const float* h, int h_broadcast, const float* norm_w, const float* w, const float* alpha, const float* beta,
                          const float* lin_w, const float* lin_b, int B, int n, int D, int sk_it, float* pre_out, float* post_out,
                          float* mix_out, float* stats_out, float* read_out, float* z_out, float* h_out, void* stream) {
  if (!h || !norm_w || !w || !alpha || !beta || !lin_w || !lin_b || !pre_out || !post_out || !mix_out || !stats_out || !read_out ||
      !z_out || !h_out || B < 0 || n != 2 || D != 128 || sk_it < 0)
    return -22;
  if (B == 0) return 0;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mhc_sub_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSubLdsBytes) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr = true;
  }
  SubFwdArgs a{h, norm_w, w, alpha, beta, lin_w, lin_b, pre_out, post_out, mix_out, stats_out, read_out, z_out, h_out, B, sk_it,
               h_broadcast ? D : 2 * D, h_broadcast ? 0 : D};
  int blocks = ((B + 15) / 16 + kSubWaves - 1) / kSubWaves;
  if (blocks > 256) blocks = 256;                          // one workgroup per CU (145 KB of LDS), its waves walk the tiles
  hipLaunchKernelGGL(mhc_sub_fwd_kernel, dim3(blocks), dim3(64 * kSubWaves), kSubLdsBytes, (hipStream_t)stream, a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_sub_backward(const float* g, int g_broadcast, const float* h, int h_broadcast, const float* z, const float* pre,
                           const float* post, const float* mix, const float* stats, const float* norm_w, const float* w,
                           const float* alpha, const float* lin_w, int B, int n, int D, float* d_z, float* d_h, int sum_branches,
                           float* d_norm_w, float* d_w, float* d_alpha, float* d_beta, void* workspace, void* stream) {
  if (!g || !h || !z || !pre || !post || !mix || !stats || !norm_w || !w || !alpha || !lin_w || !d_z || !d_h || !d_norm_w || !d_w ||
      !d_alpha || !d_beta || !workspace || B < 1 || n != 2 || D != 128)
    return -22;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mhc_sub_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSubBwdLdsBytes) != hipSuccess ||
        hipFuncSetAttribute((const void*)mhc_sub_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSubBwdLdsBytes) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr = true;
  }
  int blocks = ((B + 15) / 16 + kSubBwdWaves - 1) / kSubBwdWaves;
  if (blocks > 256) blocks = 256;                          // one workgroup per CU (110 KB of LDS), its waves walk the tiles
  SubBwdArgs a{g, h, z, pre, post, mix, stats, norm_w, w, alpha, lin_w, d_z, d_h, static_cast<float*>(workspace), B,
               g_broadcast ? D : 2 * D, g_broadcast ? 0 : D, h_broadcast ? D : 2 * D, h_broadcast ? 0 : D};
  if (sum_branches) hipLaunchKernelGGL(mhc_sub_bwd_kernel<true>, dim3(blocks), dim3(64 * kSubBwdWaves), kSubBwdLdsBytes, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mhc_sub_bwd_kernel<false>, dim3(blocks), dim3(64 * kSubBwdWaves), kSubBwdLdsBytes, (hipStream_t)stream, a);
  ReduceArgs r{static_cast<const float*>(workspace), blocks, kGatesLen, 4,
               {256, 256 + 256 * 8, 256 + 256 * 8 + 3, kGatesLen}, {256, 256 * 8, 0, 0}, {d_norm_w, d_w, d_alpha, d_beta}};
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((kGatesLen + 31) / 32, 1), dim3(256), 0, (hipStream_t)stream, r);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

#ifdef GYMRL_PROF_BUILD
int gymrl_mhc_sub_bwd_prof_read(unsigned long long* out8, int reset) {   // probe build only (not in include/gymrl.h)
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sub_bwd_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sub_bwd_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

size_t gymrl_mhc_policy_image_floats(int n_sub) { return n_sub < 0 || n_sub > kPolMaxSub ? 0 : policy_image_floats(n_sub); }

int gymrl_mhc_policy_pack(const gymrl_mhc_policy* p, float* image, void* stream) {
  if (!image || (reinterpret_cast<uintptr_t>(image) & 15)) return -22;
  PolicyArgs a{};
  gymrl_mhc_policy q;
  if (!p) return -22;
  q = *p; q.image = nullptr;
  if (const int rc = policy_fill(a, &q)) return rc;
  const size_t total = policy_image_floats(a.n_sub);
  hipLaunchKernelGGL(mhc_policy_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, image);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_policy_forward(const gymrl_mhc_policy* p, const float* obs, int B, float* logits_out, float* value_out, void* stream) {
  if (!obs || !logits_out || !value_out || B < 0) return -22;
  PolicyArgs a{};
  if (const int rc = policy_fill(a, p)) return rc;
  if (B == 0) return 0;
  hipLaunchKernelGGL(mhc_policy_kernel, dim3((B + 15) / 16), dim3(256), 0, (hipStream_t)stream, a, obs, B, logits_out, value_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
