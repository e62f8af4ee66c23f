// optim.hip — fused clip_grad_norm_ + Adam and Polyak soft update on flat fp32
// parameter buffers (gfx950).
//
//   O1  gymrl_sqnorm + gymrl_adam_step   ppo_lunarlander.py:169,302-307
//       (also dqn_cartpole.py:163-166 with clamp_abs=1, rainbow_dqn_cartpole.py:343-345,
//        sac_pendulum.py:244-263)
//   R4/A4 gymrl_soft_update              rainbow_dqn_cartpole.py:347-352, sac_pendulum.py:194-199
//
// HBM-bound: Adam moves 28 B/param (p,g,m,v read; p,m,v written) + 4 B when the
// fused zero_grad rewrites g, + 4 B for the norm pre-pass.  The whole model is one
// flat buffer so each optimiser step is two launches regardless of layer count.
// The squared norm is reduced block-partials -> fixed-order final sum (no float
// atomics) so parameters are bit-reproducible run to run.
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

using namespace gymrl;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxParts = 1024;

__global__ __launch_bounds__(kBlock) void sqnorm_partial_kernel(const float* __restrict__ g,
                                                                int64_t n, float gs,
                                                                double* __restrict__ partials) {
  double s = 0.0;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    const float a = v.x * gs, b = v.y * gs, c = v.z * gs, d = v.w * gs;
    s += (double)a * a + (double)b * b + (double)c * c + (double)d * d;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float a = g[(n4 << 2) + threadIdx.x] * gs;
    s += (double)a * a;
  }
  __shared__ double sm[kBlock / 64];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) t += sm[w];
    partials[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void sqnorm_final_kernel(const double* __restrict__ partials,
                                                              int nparts,
                                                              double* __restrict__ out) {
  __shared__ double sm[kBlock];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) a += partials[i];
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0];
}

// Up to 256 bytes of host scalars travel as a kernel argument and land in device memory: no
// staging buffer whose reuse would have to be fenced, and stream-ordered like any launch.
struct ScalarBlock { uint32_t w[64]; };
__global__ void store_scalars_kernel(uint32_t* __restrict__ dst, ScalarBlock blk, int nwords) {
  if (blockIdx.x == 0 && (int)threadIdx.x < nwords) dst[threadIdx.x] = blk.w[threadIdx.x];
}

// A chunk of vector steps replayed as one hipGraph stages every step's scalars at once: up to 3840 bytes.
struct ScalarBlockBig { uint32_t w[960]; };
__global__ void store_scalars_big_kernel(uint32_t* __restrict__ dst, ScalarBlockBig blk, int nwords) {
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) dst[i] = blk.w[i];
}

struct AdamArgs {
  float step_size_host;  // (float)(lr / (1 - beta1^t)), python-double arithmetic
  float inv_bc1;         // 1 / (1 - beta1^t)   (used with a device-resident lr)
  float bc2_sqrt;        // sqrt(1 - beta2^t)
  float omb1, beta2, omb2, eps;  // 1-beta1, beta2, 1-beta2 rounded from double like torch
  float grad_scale, max_grad_norm, clamp_abs;
  int zero_grad;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, const AdamArgs& a,
                                         float scale, float step_size) {
  float gg = g * a.grad_scale;
  gg = gg * scale;
  if (a.clamp_abs > 0.0f) gg = fminf(fmaxf(gg, -a.clamp_abs), a.clamp_abs);
  m = m + (gg - m) * a.omb1;                // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + a.omb2 * gg * gg;        // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - step_size * (m / denom);                     // addcdiv_(exp_avg, denom, -step_size)
  if (a.zero_grad) g = 0.0f;
}

template <bool FOLD>      // FOLD: gymrl_clip_adam_step's form (the plain step carries neither the shared array nor the fold)
__global__ __launch_bounds__(kBlock) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      int64_t n, AdamArgs a,
                                                      const float* __restrict__ lr_dev,
                                                      const float* __restrict__ bias_dev,
                                                      const double* __restrict__ sqnorm,
                                                      float* __restrict__ polyak, float tau, float omt,
                                                      const double* __restrict__ partials, int nparts,
                                                      double* __restrict__ sqnorm_out) {
  if (bias_dev) {          // graph replay: the step-dependent scalars live in device memory
    a.step_size_host = bias_dev[0]; a.inv_bc1 = bias_dev[1]; a.bc2_sqrt = bias_dev[2];
  }
  // gymrl_clip_adam_step: the squared norm's second level here instead of in a launch of its own — every workgroup folds
  // sqnorm_partial_kernel's block partials exactly as sqnorm_final_kernel does (the same bits in every workgroup)
  double folded = 0.0;
  if constexpr (FOLD) {
    __shared__ double fold[kBlock];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) acc += partials[i];
    fold[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) fold[threadIdx.x] += fold[threadIdx.x + s];
      __syncthreads();
    }
    folded = fold[0];
    if (sqnorm_out && blockIdx.x == 0 && threadIdx.x == 0) sqnorm_out[0] = folded;
    sqnorm = &folded;
  }
  float scale = 1.0f;
  if (a.max_grad_norm > 0.0f && sqnorm) {
    const float total = (float)sqrt(sqnorm[0]);
    const float coef = a.max_grad_norm / (total + 1e-6f);
    scale = coef < 1.0f ? coef : 1.0f;
  }
  const float step_size = lr_dev ? lr_dev[0] * a.inv_bc1 : a.step_size_host;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    adam_one(P.x, G.x, M.x, V.x, a, scale, step_size);
    adam_one(P.y, G.y, M.y, V.y, a, scale, step_size);
    adam_one(P.z, G.z, M.z, V.z, a, scale, step_size);
    adam_one(P.w, G.w, M.w, V.w, a, scale, step_size);
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
    if (a.zero_grad) reinterpret_cast<float4*>(g)[i] = G;
    if (polyak) {            // the soft target update of the parameters just written (soft_update_kernel's expression)
      float4 t = reinterpret_cast<float4*>(polyak)[i];
      t.x = tau * P.x + omt * t.x; t.y = tau * P.y + omt * t.y;
      t.z = tau * P.z + omt * t.z; t.w = tau * P.w + omt * t.w;
      reinterpret_cast<float4*>(polyak)[i] = t;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float P = p[i], G = g[i], M = m[i], V = v[i];
    adam_one(P, G, M, V, a, scale, step_size);
    p[i] = P; m[i] = M; v[i] = V;
    if (a.zero_grad) g[i] = G;
    if (polyak) polyak[i] = tau * P + omt * polyak[i];
  }
}

__global__ __launch_bounds__(kBlock) void soft_update_kernel(float* __restrict__ tgt,
                                                             const float* __restrict__ src,
                                                             int64_t n, float tau, float omt) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 t = reinterpret_cast<float4*>(tgt)[i];
    const float4 s = reinterpret_cast<const float4*>(src)[i];
    t.x = tau * s.x + omt * t.x; t.y = tau * s.y + omt * t.y;
    t.z = tau * s.z + omt * t.z; t.w = tau * s.w + omt * t.w;
    reinterpret_cast<float4*>(tgt)[i] = t;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    tgt[i] = tau * src[i] + omt * tgt[i];
  }
}

inline int grid_for(int64_t n, int per_thread) {
  int64_t nb = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  if (nb < 1) nb = 1;
  if (nb > kMaxParts) nb = kMaxParts;
  return (int)nb;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

size_t gymrl_reduce_workspace_bytes(void) { return sizeof(double) * 16384; }

int gymrl_sqnorm(const float* g, int64_t n, float grad_scale, double* sqnorm_out, void* workspace,
                 void* stream_) {
  if (!g || !sqnorm_out || !workspace || n < 0 || !aligned16(g)) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(n, 16);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(kBlock), 0, stream, g, n, grad_scale,
                     (double*)workspace);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(kBlock), 0, stream,
                     (const double*)workspace, nb, sqnorm_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

// Host arithmetic of the step-dependent scalars, exactly as gymrl_adam_step does it.
int gymrl_adam_bias(double lr, double beta1, double beta2, int64_t step, float* out_host) {
  if (!out_host || step < 1) return -22;
  const double bc1 = 1.0 - __builtin_pow(beta1, (double)step);
  const double bc2 = 1.0 - __builtin_pow(beta2, (double)step);
  out_host[0] = (float)(lr / bc1);
  out_host[1] = (float)(1.0 / bc1);
  out_host[2] = (float)__builtin_sqrt(bc2);
  out_host[3] = 0.0f;
  return 0;
}

int gymrl_store_scalars(void* dst_dev, const void* src_host, int nbytes, void* stream_) {
  if (!dst_dev || !src_host || nbytes <= 0 || nbytes > (int)sizeof(ScalarBlockBig) || (nbytes & 3) ||
      (reinterpret_cast<uintptr_t>(dst_dev) & 3))
    return -22;
  if (nbytes > (int)sizeof(ScalarBlock)) {
    ScalarBlockBig big;
    __builtin_memcpy(big.w, src_host, (size_t)nbytes);
    hipLaunchKernelGGL(store_scalars_big_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, (uint32_t*)dst_dev, big,
                       nbytes >> 2);
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  ScalarBlock blk;
  __builtin_memcpy(blk.w, src_host, (size_t)nbytes);
  hipLaunchKernelGGL(store_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, (uint32_t*)dst_dev, blk,
                     nbytes >> 2);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_adam_step(float* p, float* g, float* m, float* v, int64_t n, double lr_host,
                    const float* lr_dev, double beta1, double beta2, double eps, int64_t step,
                    const float* bias_dev, float grad_scale, float max_grad_norm, const double* sqnorm,
                    float clamp_abs, int zero_grad, float* polyak_target, double tau, void* stream_) {
  if (!p || !g || !m || !v || n < 0 || (step < 1 && !bias_dev) || (polyak_target && !aligned16(polyak_target))) return -22;
  if (bias_dev) step = 1;   // unused: the kernel takes step_size / bc2_sqrt from bias_dev
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return -22;
  if (max_grad_norm > 0.0f && !sqnorm) return -22;
  if (n == 0) return 0;
  AdamArgs a;
  // bias corrections as torch.optim.Adam computes them: python doubles, then f32.
  const double bc1 = 1.0 - __builtin_pow(beta1, (double)step);
  const double bc2 = 1.0 - __builtin_pow(beta2, (double)step);
  a.step_size_host = (float)(lr_host / bc1);
  a.inv_bc1 = (float)(1.0 / bc1);
  a.bc2_sqrt = (float)__builtin_sqrt(bc2);
  a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.grad_scale = grad_scale; a.max_grad_norm = max_grad_norm; a.clamp_abs = clamp_abs;
  a.zero_grad = zero_grad;
  hipLaunchKernelGGL(adam_kernel<false>, dim3(grid_for(n, 4)), dim3(kBlock), 0, (hipStream_t)stream_, p, g,
                     m, v, n, a, lr_dev, bias_dev, sqnorm, polyak_target, (float)tau, (float)(1.0 - tau),
                     (const double*)nullptr, 0, (double*)nullptr);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

// clip_grad_norm_ + Adam as TWO launches instead of three: gymrl_sqnorm's first level, then gymrl_adam_step's kernel with the
// norm's second level folded in (above).  Same sums in the same order: the same bits.
int gymrl_clip_adam_step(float* p, float* g, float* m, float* v, int64_t n, double lr_host, const float* lr_dev, double beta1,
                         double beta2, double eps, int64_t step, const float* bias_dev, float grad_scale, float max_grad_norm,
                         double* sqnorm_out, float clamp_abs, int zero_grad, float* polyak_target, double tau, void* workspace,
                         void* stream_) {
  if (!p || !g || !m || !v || !workspace || n < 0 || (step < 1 && !bias_dev) || (polyak_target && !aligned16(polyak_target)) ||
      !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) || !(max_grad_norm > 0.0f))
    return -22;
  if (bias_dev) step = 1;
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) {                 // no parameters: the norm gymrl_sqnorm would have written is 0 (a reader of sqnorm_out never sees a stale one)
    if (sqnorm_out && hipMemsetAsync(sqnorm_out, 0, sizeof(double), stream) != hipSuccess) return -5;
    return 0;
  }
  AdamArgs a;
  const double bc1 = 1.0 - __builtin_pow(beta1, (double)step);
  const double bc2 = 1.0 - __builtin_pow(beta2, (double)step);
  a.step_size_host = (float)(lr_host / bc1);
  a.inv_bc1 = (float)(1.0 / bc1);
  a.bc2_sqrt = (float)__builtin_sqrt(bc2);
  a.omb1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.omb2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.grad_scale = grad_scale; a.max_grad_norm = max_grad_norm; a.clamp_abs = clamp_abs;
  a.zero_grad = zero_grad;
  const int nb = grid_for(n, 16);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(kBlock), 0, stream, g, n, grad_scale, (double*)workspace);
  hipLaunchKernelGGL(adam_kernel<true>, dim3(grid_for(n, 4)), dim3(kBlock), 0, stream, p, g, m, v, n, a, lr_dev, bias_dev,
                     (const double*)nullptr, polyak_target, (float)tau, (float)(1.0 - tau), (const double*)workspace, nb, sqnorm_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_soft_update(float* target, const float* source, int64_t n, double tau, void* stream_) {
  if (!target || !source || n < 0 || !aligned16(target) || !aligned16(source)) return -22;
  if (n == 0) return 0;
  hipLaunchKernelGGL(soft_update_kernel, dim3(grid_for(n, 4)), dim3(kBlock), 0,
                     (hipStream_t)stream_, target, source, n, (float)tau, (float)(1.0 - tau));
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
