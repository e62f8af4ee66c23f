// mhc.hip — inference-side pieces of PPO-full's manifold-hyper-connection backbone (rollout forward only).
//
// ppo_full_lunarlander.py:106-250: every MHCBlock half reads the branch stack h [B, n, D] through per-sample gates —
// an RMS-fused linear read-out (n*D -> n*n + 2n numbers per row), two sigmoids, an exp and `max_sk_it` Sinkhorn-Knopp
// sweeps on an n x n matrix — mixes the branches, runs ONE D x D Linear + SiLU on the weighted branch sum and writes
// the stack back.  Through PyTorch that is ~95 launches per half (each Sinkhorn sweep alone is 6), ~400 per rollout
// forward at 4096 rows: 3 ms per vector step of pure launch cost, 12 of the 47 s of a config-5 iteration.  For the
// rollout (no gradients) a half is three launches here:
//
//   gymrl_mhc_gates     h -> pre [B, n], post [B, n], mix [B, n, n] and read [B, D] = sum_i pre_i h_i   (one wave per row)
//   gymrl_lin_fwd       out = SiLU(read W^T + b)                                                        (csrc/lin.hip)
//   gymrl_mhc_combine   h'[b, i, :] = post_i out + sum_j mix_ij h[b, j, :]
//
// plus gymrl_rmsnorm (optionally over the branch sum: MHCBackbone.final_norm(h.sum(1)), and the RMSNorm of the actor /
// critic MLPs).  The training pass keeps the torch modules (SURVEY 8a F1 leaves the network to PyTorch-ROCm); these
// kernels are compared with them at 1e-5 (tests/test_mhc_fused_gpu.py).
#include "train_device.hpp"
#include "../../include/gymrl.h"

namespace {

using namespace gymrl;

constexpr int kWaves = 4;
constexpr int kMaxN = 4;                         // branches (mhc_rate)

struct GatesArgs {
  const float* h; const float* norm_w; const float* w; const float* alpha; const float* beta;
  float* pre; float* post; float* mix; float* read;
  int B, D, sk_it;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

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
    for (int j = 0; j < N; ++j) A[i][j] = expf(r_inv * Hs[2 * N + i * N + j] * a2 + a.beta[2 * N + i * N + j]);
  }
  for (int it = 0; it < a.sk_it; ++it) {                   // Sinkhorn-Knopp scalings (:141-146)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < N; ++j) s += A[i][j] * v[j];
      u[i] = 1.0f / (s + 1e-8f);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) s += A[i][j] * u[i];
      v[j] = 1.0f / (s + 1e-8f);
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

template <int N>
__global__ __launch_bounds__(256) void mhc_combine_kernel(const float* __restrict__ post, const float* __restrict__ mix,
                                                          const float* __restrict__ out, const float* __restrict__ h, int B,
                                                          int D, float* __restrict__ h_out) {
  const int64_t total = (int64_t)B * D;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t b = t / D;
    const int d = (int)(t % D);
    float hv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) hv[j] = h[(b * N + j) * D + d];
    const float o = out[b * D + d];
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

// one wave per row: d_pre[b, i] = sum_d g[b, d] h[b, i, d];  d_h[b, i, d] (+)= pre[b, i] g[b, d]
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
      d_h[o] = accumulate ? d_h[o] + p[i] * gv : p[i] * gv;
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
                                                                    const float* __restrict__ h, int B, int D,
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
    const float o = out[row * D + d];
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
    d_out[row * D + d] = so;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float sh = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) sh += mx[i][j] * gv[i];
      d_h[(row * N + j) * D + d] = sh;
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
// mix = u exp(z[2n:]) v with u, v constants (the reference computes them under no_grad).  Given dL/d(pre, post, mix) one wave per
// row recomputes H and r, forms dz (sigmoid' / exp' from the saved outputs), dH = dz r alpha, d r = sum dz H alpha, and in a second
// pass over the row writes d flat = norm_w (dH w^T) + d|flat| flat / |flat|; the parameter gradients (d norm_w, d w, d alpha,
// d beta: sums over rows) are accumulated in registers over the rows a wave visits, added across the workgroup's waves through
// LDS in a fixed order, and written as one partial vector per workgroup for gates_bwd_reduce_kernel: no atomics.
struct GatesBwdArgs {
  const float* h; const float* norm_w; const float* w; const float* alpha;
  const float* pre; const float* post; const float* mix;
  const float* d_pre; const float* d_post; const float* d_mix;
  float* d_h; float* partial;
  int B, D;
};

template <int CH>        // nc = 256 * CH columns, n = 2
__global__ __launch_bounds__(64 * kWaves) void mhc_gates_bwd_kernel(const GatesBwdArgs a) {
  constexpr int N = 2, G = N * N + 2 * N;
  extern __shared__ float red[];                           // [kWaves][len]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = 256 * CH, len = nc + nc * G + 3 + G;
  f32x4 nw[CH];
  float wr[CH][4][G];                                      // this lane's rows of w: constants of the launch
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int c = 256 * ch + 4 * lane;
    nw[ch] = *reinterpret_cast<const f32x4*>(a.norm_w + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < G; ++k) wr[ch][e][k] = a.w[(size_t)(c + e) * G + k];
  }
  const float al[3] = {a.alpha[0], a.alpha[1], a.alpha[2]};
  float acc_nw[CH][4], acc_w[CH][4][G], acc_al[3] = {0.0f, 0.0f, 0.0f}, acc_be[G];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc_nw[ch][e] = 0.0f;
#pragma unroll
      for (int k = 0; k < G; ++k) acc_w[ch][e][k] = 0.0f;
    }
#pragma unroll
  for (int k = 0; k < G; ++k) acc_be[k] = 0.0f;
  const float inv_sqrt_nc = 1.0f / sqrtf((float)nc);
  for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < a.B; row += (int64_t)gridDim.x * kWaves) {
    const float* hr = a.h + row * nc;
    f32x4 x[CH];
    float Hs[G], sq = 0.0f;
#pragma unroll
    for (int k = 0; k < G; ++k) Hs[k] = 0.0f;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      x[ch] = *reinterpret_cast<const f32x4*>(hr + 256 * ch + 4 * lane);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = nw[ch][e] * x[ch][e];
        sq += x[ch][e] * x[ch][e];
#pragma unroll
        for (int k = 0; k < G; ++k) Hs[k] += t * wr[ch][e][k];
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sq += __shfl_xor(sq, off, 64);
#pragma unroll
      for (int k = 0; k < G; ++k) Hs[k] += __shfl_xor(Hs[k], off, 64);
    }
    const float norm = sqrtf(sq);
    const float r = 1.0f / (norm * inv_sqrt_nc + 1e-6f);
    float dz[G];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float p = a.pre[row * N + i], q = a.post[row * N + i];
      dz[i] = a.d_pre[row * N + i] * p * (1.0f - p);
      dz[N + i] = a.d_post[row * N + i] * q * (1.0f - 0.5f * q);
#pragma unroll
      for (int j = 0; j < N; ++j) dz[2 * N + i * N + j] = a.d_mix[(row * N + i) * N + j] * a.mix[(row * N + i) * N + j];
    }
    float dH[G], d_r = 0.0f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const float ag = al[k < N ? 0 : (k < 2 * N ? 1 : 2)];
      dH[k] = dz[k] * r * ag;
      d_r += dz[k] * Hs[k] * ag;
      acc_al[k < N ? 0 : (k < 2 * N ? 1 : 2)] += dz[k] * r * Hs[k];
      acc_be[k] += dz[k];
    }
    const float d_norm = d_r * (-r * r * inv_sqrt_nc);
    const float dn_over = norm > 0.0f ? d_norm / norm : 0.0f;
    float* dhr = a.d_h + row * nc;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      f32x4 dx;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t2 = 0.0f;
#pragma unroll
        for (int k = 0; k < G; ++k) t2 += dH[k] * wr[ch][e][k];
        dx[e] = nw[ch][e] * t2 + dn_over * x[ch][e];
        acc_nw[ch][e] += x[ch][e] * t2;
        const float t = nw[ch][e] * x[ch][e];
#pragma unroll
        for (int k = 0; k < G; ++k) acc_w[ch][e][k] += t * dH[k];
      }
      *reinterpret_cast<f32x4*>(dhr + 256 * ch + 4 * lane) = dx;
    }
  }
  // workgroup reduction in a fixed order, one partial vector per workgroup
  float* mine = red + (size_t)wave * len;
#pragma unroll
  for (int ch = 0; ch < CH; ++ch)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 256 * ch + 4 * lane + e;
      mine[c] = acc_nw[ch][e];
#pragma unroll
      for (int k = 0; k < G; ++k) mine[nc + c * G + k] = acc_w[ch][e][k];
    }
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) mine[nc + nc * G + g] = acc_al[g];
#pragma unroll
    for (int k = 0; k < G; ++k) mine[nc + nc * G + 3 + k] = acc_be[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < len; i += 64 * kWaves) {
    float sum = red[i];
#pragma unroll
    for (int w2 = 1; w2 < kWaves; ++w2) sum += red[(size_t)w2 * len + i];
    a.partial[(size_t)blockIdx.x * len + i] = sum;
  }
}

// out[i] = sum over workgroups (ascending) of partial[b][i], scattered to the four gradient tensors
__global__ __launch_bounds__(256) void gates_bwd_reduce_kernel(const float* __restrict__ partial, int blocks, int nc, int G,
                                                              float* __restrict__ d_nw, float* __restrict__ d_w,
                                                              float* __restrict__ d_alpha, float* __restrict__ d_beta) {
  const int len = nc + nc * G + 3 + G;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  float s = 0.0f;
  for (int b = 0; b < blocks; ++b) s += partial[(size_t)b * len + i];
  if (i < nc) d_nw[i] = s;
  else if (i < nc + nc * G) d_w[i - nc] = s;
  else if (i < nc + nc * G + 3) d_alpha[i - nc - nc * G] = s;
  else d_beta[i - nc - nc * G - 3] = s;
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
      u[i] = 1.0f / (s + 1e-8f);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) s += a[i][j] * u[i];
      v[j] = 1.0f / (s + 1e-8f);
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) { u_out[(size_t)b * N + i] = u[i]; v_out[(size_t)b * N + i] = v[i]; }
}

// y = x * rsqrt(mean(x^2) + eps) * w per row; n_sum > 1: x = the sum of n_sum consecutive [D] blocks of the row
__global__ __launch_bounds__(64 * kWaves) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             int B, int D, int n_sum, float eps, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + (size_t)row * n_sum * D;
  float sq = 0.0f;
  for (int d = lane; d < D; d += 64) {
    float s = xr[d];
    for (int k = 1; k < n_sum; ++k) s += xr[k * D + d];
    sq += s * s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  const float r = rsqrtf(sq / (float)D + eps);
  for (int d = lane; d < D; d += 64) {
    float s = xr[d];
    for (int k = 1; k < n_sum; ++k) s += xr[k * D + d];
    y[(size_t)row * D + d] = s * r * w[d];
  }
}

}  // namespace

extern "C" {

int gymrl_mhc_gates(const float* h, const float* norm_w, const float* w, const float* alpha, const float* beta, int B, int n,
                    int D, int sk_it, float* pre_out, float* post_out, float* mix_out, float* read_out, void* stream) {
  if (!h || !norm_w || !w || !alpha || !beta || !pre_out || !post_out || !mix_out || !read_out || B < 0 || D < 4 || D % 4 ||
      sk_it < 0 || (n != 2 && n != 4))
    return -22;
  if (B == 0) return 0;
  GatesArgs a{h, norm_w, w, alpha, beta, pre_out, post_out, mix_out, read_out, B, D, sk_it};
  const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
  if (n == 2) hipLaunchKernelGGL(mhc_gates_kernel<2>, grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mhc_gates_kernel<4>, grid, block, 0, (hipStream_t)stream, a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_combine(const float* post, const float* mix, const float* out, const float* h, int B, int n, int D, float* h_out,
                      void* stream) {
  if (!post || !mix || !out || !h || !h_out || B < 0 || D < 1 || (n != 2 && n != 4)) return -22;
  if (B == 0) return 0;
  int64_t nb = ((int64_t)B * D + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (n == 2) hipLaunchKernelGGL(mhc_combine_kernel<2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, post, mix, out, h, B, D, h_out);
  else hipLaunchKernelGGL(mhc_combine_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, post, mix, out, h, B, D, h_out);
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
  if (!g || !pre || !h || !d_pre || !d_h || B < 0 || D < 1 || (n != 2 && n != 4)) return -22;
  if (B == 0) return 0;
  const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
  if (n == 2) hipLaunchKernelGGL(mhc_read_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g, pre, h, B, D, d_pre, d_h, accumulate);
  else hipLaunchKernelGGL(mhc_read_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, g, pre, h, B, D, d_pre, d_h, accumulate);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mhc_combine_bwd(const float* g, const float* post, const float* mix, const float* out, const float* h, int B, int n, int D,
                          float* d_post, float* d_mix, float* d_out, float* d_h, void* stream) {
  if (!g || !post || !mix || !out || !h || !d_post || !d_mix || !d_out || !d_h || B < 0 || D < 1 || (n != 2 && n != 4)) return -22;
  if (B == 0) return 0;
  const dim3 grid((B + kWaves - 1) / kWaves), block(64 * kWaves);
  if (n == 2) hipLaunchKernelGGL(mhc_combine_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, g, post, mix, out, h, B, D, d_post, d_mix, d_out, d_h);
  else hipLaunchKernelGGL(mhc_combine_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, g, post, mix, out, h, B, D, d_post, d_mix, d_out, d_h);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

size_t gymrl_mhc_gates_bwd_workspace_bytes(int n, int D) {
  const int nc = n * D, G = n * n + 2 * n;
  return sizeof(float) * 1024 * ((size_t)nc + (size_t)nc * G + 3 + G);
}

int gymrl_mhc_gates_bwd(const float* h, const float* norm_w, const float* w, const float* alpha, const float* pre, const float* post,
                        const float* mix, const float* d_pre, const float* d_post, const float* d_mix, int B, int n, int D,
                        float* d_h, float* d_norm_w, float* d_w, float* d_alpha, float* d_beta, void* workspace, void* stream) {
  if (!h || !norm_w || !w || !alpha || !pre || !post || !mix || !d_pre || !d_post || !d_mix || !d_h || !d_norm_w || !d_w ||
      !d_alpha || !d_beta || !workspace || B < 1 || n != 2 || (n * D != 256 && n * D != 512))
    return -22;
  const int nc = n * D, G = n * n + 2 * n, len = nc + nc * G + 3 + G;
  int blocks = (B + kWaves - 1) / kWaves;
  if (blocks > 1024) blocks = 1024;
  GatesBwdArgs a{h, norm_w, w, alpha, pre, post, mix, d_pre, d_post, d_mix, d_h, static_cast<float*>(workspace), B, D};
  const size_t lds = sizeof(float) * kWaves * len;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)mhc_gates_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr = true;
  }
  if (nc == 256) hipLaunchKernelGGL(mhc_gates_bwd_kernel<1>, dim3(blocks), dim3(64 * kWaves), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mhc_gates_bwd_kernel<2>, dim3(blocks), dim3(64 * kWaves), lds, (hipStream_t)stream, a);
  hipLaunchKernelGGL(gates_bwd_reduce_kernel, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const float*>(workspace), blocks, nc, G, d_norm_w, d_w, d_alpha, d_beta);
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

int gymrl_rmsnorm(const float* x, const float* w, int B, int D, int n_sum, float eps, float* y, void* stream) {
  if (!x || !w || !y || B < 0 || D < 1 || n_sum < 1) return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((B + kWaves - 1) / kWaves), dim3(64 * kWaves), 0, (hipStream_t)stream, x, w, B, D, n_sum,
                     eps, y);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
