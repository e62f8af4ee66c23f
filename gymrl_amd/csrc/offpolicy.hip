// offpolicy.hip — DQN / Rainbow / SAC learner-side kernels and running normalisation.
//
//   R1  gymrl_noisy_noise        rainbow_dqn_cartpole.py:77-87
//   D3  gymrl_epsilon_greedy     dqn_cartpole.py:117-133
//   D4/R4 gymrl_dqn_td_loss      dqn_cartpole.py:157-161, rainbow_dqn_cartpole.py:319-338,
//                                ddqn_per_cartpole.py:224-233
//   A1  gymrl_sac_sample_fwd/bwd sac_pendulum.py:76-87
//   A4  gymrl_sac_target / _critic_loss / _actor_loss / _alpha_step   sac_pendulum.py:233-263
//   N1-N3 gymrl_running_norm / gymrl_reward_scaling   utils/normalization.py:4-52
//   8f.3 gymrl_noisy_action / gymrl_mse_loss / gymrl_neg_mean_loss   ddpg_pendulum.py:143-185, td3_pendulum.py:164-213
//
// All of these are per-sample elementwise maps over B <= a few thousand rows of <= 8
// words: launch-latency bound at the reference batch sizes, HBM-streaming at large B.
// Loss sums are reduced block-partials -> fixed-order final sum (no float atomics).
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

using namespace gymrl;

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int grid_for(int n) { int nb = cdiv(n, kBlock); return nb < 1 ? 1 : (nb > kMaxBlocks ? kMaxBlocks : nb); }

template <int K>
__device__ __forceinline__ void block_partials(double (&v)[K], double* __restrict__ partials,
                                               double* __restrict__ direct = nullptr) {
  __shared__ double sm[K][kBlock / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = wave_sum(v[k]);
    if (lane == 0) sm[k][wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) s += sm[threadIdx.x][w];
    // a single-block launch adds to the destination itself (what finalize_kernel would do with its one partial,
    // bit for bit) and the finalize launch is skipped
    if (direct) direct[threadIdx.x] += s;
    else partials[(size_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// out[off + k] += sum_blocks partials[block][k]
template <int K>
__global__ __launch_bounds__(kBlock) void finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                          double* __restrict__ out) {
  __shared__ double sm[K][kBlock / 64];
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += partials[(size_t)i * K + k];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = wave_sum(v[k]);
    if (lane == 0) sm[k][wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) s += sm[threadIdx.x][w];
    out[threadIdx.x] += s;
  }
}

// ------------------------------------------------------------------ R1 -------
// scale_noise / box_muller: gymrl_device.hpp (shared with the fused NoisyLinear head, lin.hip)
__global__ __launch_bounds__(kBlock) void noisy_noise_kernel(const float* __restrict__ eps_in,
                                                             const float* __restrict__ eps_out,
                                                             uint64_t seed, uint64_t counter, int nin,
                                                             int nout, float* __restrict__ w_eps,
                                                             float* __restrict__ b_eps,
                                                             const uint64_t* __restrict__ counter_dev) {
  if (counter_dev) counter = counter_dev[0];   // recorded into a hipGraph: this replay's draw counter lives on the device
  const int64_t total = (int64_t)nin * nout;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int j = (int)(t / nin), i = (int)(t % nin);
    const float ei = eps_in ? eps_in[i] : box_muller(seed, counter, 0u, (uint32_t)i);
    const float ej = eps_out ? eps_out[j] : box_muller(seed, counter, 1u, (uint32_t)j);
    const float fj = scale_noise(ej);
    w_eps[t] = fj * scale_noise(ei);                                 // torch.outer(epsilon_j, epsilon_i)
    if (i == 0) b_eps[j] = fj;
  }
}

// ------------------------------------------------------------------ D3 -------
__global__ __launch_bounds__(kBlock) void epsilon_greedy_kernel(const float* __restrict__ q,
                                                                const float* __restrict__ u, uint64_t seed,
                                                                uint64_t counter, int64_t env_id0, int n,
                                                                int A, float epsilon,
                                                                int32_t* __restrict__ act) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float u0, u1;
  if (u) { u0 = u[2 * i]; u1 = u[2 * i + 1]; }
  else {
    const uint64_t env = (uint64_t)(env_id0 + i);
    const u32x4 r = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), (uint32_t)counter,
                               RNG_POLICY | 0x08000000u | (uint32_t)((counter >> 32) & 0x07FFFFFFu));
    u0 = u01f(r.x); u1 = u01f(r.y);
  }
  int a;
  if (u0 < epsilon) {                                                // random.random() < eps -> action_space.sample()
    a = (int)(u1 * (float)A);
    a = a >= A ? A - 1 : a;
  } else {
    a = 0;
    float best = q[(size_t)i * A];
    for (int k = 1; k < A; ++k) { const float v = q[(size_t)i * A + k]; if (v > best) { best = v; a = k; } }
  }
  act[i] = a;
}

// --------------------------------------------------------------- D4 / R4 -----
__global__ __launch_bounds__(kBlock) void dqn_td_kernel(
    const float* __restrict__ q, const float* __restrict__ qn_online, const float* __restrict__ qn_target,
    const int32_t* __restrict__ act, const float* __restrict__ rew, const float* __restrict__ flag,
    const float* __restrict__ w, int B, int A, float gamma_n, float* __restrict__ td_out,
    float* __restrict__ dq_out, double* __restrict__ partials, double* __restrict__ direct) {
  double acc[1] = {0.0};
  const float invB = 1.0f / (float)B;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    const float* sel = qn_online ? qn_online + (size_t)b * A : qn_target + (size_t)b * A;
    int astar = 0;
    float best = sel[0];
    for (int k = 1; k < A; ++k) if (sel[k] > best) { best = sel[k]; astar = k; }
    const float nq = qn_target[(size_t)b * A + astar];
    const float y = rew[b] + gamma_n * nq * (1.0f - flag[b]);
    const int a = act[b];
    const float td = q[(size_t)b * A + a] - y;
    const float wb = w ? w[b] : 1.0f;
    td_out[b] = td;
    for (int k = 0; k < A; ++k) dq_out[(size_t)b * A + k] = (k == a) ? (2.0f * td) * wb * invB : 0.0f;
    acc[0] += (double)((td * td) * wb);
  }
  if (partials) block_partials<1>(acc, partials, direct);
}

// ------------------------------------------------------------------ A1 -------
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;   // math.log(math.sqrt(2*math.pi))

__global__ __launch_bounds__(kBlock) void sac_sample_fwd_kernel(const float* __restrict__ mean,
                                                                const float* __restrict__ log_std,
                                                                const float* __restrict__ eps, int B, int A,
                                                                float bound, float* __restrict__ action,
                                                                float* __restrict__ logp) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  float lp = 0.0f;
  for (int j = 0; j < A; ++j) {
    const size_t o = (size_t)b * A + j;
    const float mu = mean[o], std = det_expf(log_std[o]);
    const float x = mu + std * eps[o];                               // normal.rsample()
    const float t = det_tanhf(x);
    action[o] = t * bound;
    const float var = std * std, log_scale = det_logf(std);
    float l = -((x - mu) * (x - mu)) / (2.0f * var) - log_scale - kLogSqrt2Pi;   // Normal.log_prob
    l -= det_logf(bound * (1.0f - t * t) + 1e-6f);
    lp += l;
  }
  logp[b] = lp;
}

__global__ __launch_bounds__(kBlock) void sac_sample_bwd_kernel(const float* __restrict__ mean,
                                                                const float* __restrict__ log_std,
                                                                const float* __restrict__ eps,
                                                                const float* __restrict__ d_action,
                                                                const float* __restrict__ d_logp, int B,
                                                                int A, float bound,
                                                                float* __restrict__ d_mean,
                                                                float* __restrict__ d_log_std) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const float gl = d_logp ? d_logp[b] : 0.0f;
  for (int j = 0; j < A; ++j) {
    const size_t o = (size_t)b * A + j;
    const float mu = mean[o], std = det_expf(log_std[o]), e = eps[o];
    const float x = mu + std * e;
    const float t = det_tanhf(x);
    const float omt = 1.0f - t * t;
    const float ga = d_action ? d_action[o] : 0.0f;
    // d/dx: action = bound*tanh(x);  -log(bound*(1-t^2)+1e-6) -> +2*t*bound*(1-t^2)/(bound*(1-t^2)+1e-6)
    const float dx = ga * bound * omt + gl * (2.0f * t * bound * omt / (bound * omt + 1e-6f));
    d_mean[o] = dx;
    // x = mu + exp(ls)*eps; the Gaussian term reduces to -ls (its (x-mu)^2/(2 var) part is constant)
    d_log_std[o] = dx * (std * e) - gl;
  }
}

// ------------------------------------------------------------------ A4 -------
__global__ __launch_bounds__(kBlock) void sac_target_kernel(const float* __restrict__ rew,
                                                            const float* __restrict__ done,
                                                            const float* __restrict__ q1n,
                                                            const float* __restrict__ q2n,
                                                            const float* __restrict__ logp_n,
                                                            const double* __restrict__ log_alpha, int B,
                                                            float gamma, float* __restrict__ y) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const float alpha = (float)exp(log_alpha[0]);
  const float tq = fminf(q1n[b], q2n[b]) - alpha * logp_n[b];
  y[b] = rew[b] + gamma * (1.0f - done[b]) * tq;
}

__global__ __launch_bounds__(kBlock) void sac_critic_kernel(const float* __restrict__ q1,
                                                            const float* __restrict__ q2,
                                                            const float* __restrict__ y, int B,
                                                            float* __restrict__ dq1, float* __restrict__ dq2,
                                                            double* __restrict__ partials, double* __restrict__ direct) {
  double acc[1] = {0.0};
  const float invB = 1.0f / (float)B;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    const float e1 = q1[b] - y[b], e2 = q2[b] - y[b];
    dq1[b] = 2.0f * e1 * invB; dq2[b] = 2.0f * e2 * invB;
    acc[0] += (double)(e1 * e1) + (double)(e2 * e2);
  }
  block_partials<1>(acc, partials, direct);
}

__global__ __launch_bounds__(kBlock) void sac_actor_kernel(const float* __restrict__ logp,
                                                           const float* __restrict__ q1,
                                                           const float* __restrict__ q2,
                                                           const double* __restrict__ log_alpha, int B,
                                                           float target_entropy, float* __restrict__ dlogp,
                                                           float* __restrict__ dq1, float* __restrict__ dq2,
                                                           double* __restrict__ partials, double* __restrict__ direct) {
  double acc[2] = {0.0, 0.0};
  const float invB = 1.0f / (float)B;
  const float alpha = (float)exp(log_alpha[0]);
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    const float a = q1[b], c = q2[b];
    const float w1 = a < c ? 1.0f : (a == c ? 0.5f : 0.0f);          // torch.min tie rule
    dlogp[b] = alpha * invB;
    dq1[b] = -w1 * invB; dq2[b] = -(1.0f - w1) * invB;
    acc[0] += (double)(alpha * logp[b] - fminf(a, c));
    acc[1] += (double)(logp[b] + target_entropy);
  }
  block_partials<2>(acc, partials, direct);
}

__global__ void sac_alpha_step_kernel(double* __restrict__ log_alpha, double* __restrict__ m,
                                      double* __restrict__ v, const double* __restrict__ sums, int B,
                                      double lr, double beta1, double beta2, double eps, double bc1,
                                      double bc2_sqrt, const double* __restrict__ bias_dev,
                                      double* __restrict__ loss_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (bias_dev) { bc1 = bias_dev[0]; bc2_sqrt = sqrt(bias_dev[1]); }     // graph replay
  // alpha_loss = -(log_alpha * (logp + target_entropy).detach()).mean()
  const double mean_term = sums[2] / (double)B;
  if (loss_out) loss_out[0] = -(log_alpha[0] * mean_term);
  const double g = -mean_term;
  m[0] = m[0] + (g - m[0]) * (1.0 - beta1);
  v[0] = v[0] * beta2 + (1.0 - beta2) * g * g;
  const double denom = sqrt(v[0]) / bc2_sqrt + eps;
  log_alpha[0] = log_alpha[0] - (lr / bc1) * (m[0] / denom);
}

// ---------------------------------------------------------------- N1-N3 ------
// One lane per feature; the N rows are consumed in order (the reference's single stream).
__global__ void running_norm_kernel(const float* __restrict__ x, int N, int D, double* __restrict__ stats,
                                    int update, float* __restrict__ y) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = k < D;
  double n = stats[0];
  float mean = valid ? (float)stats[2 + k] : 0.0f;
  double S = valid ? stats[2 + D + k] : 0.0;
  double std = valid ? stats[2 + 2 * D + k] : 0.0;
  for (int i = 0; valid && i < N; ++i) {
    const float xv = x[(size_t)i * D + k];
    if (update) {
      n += 1.0;
      if (n == 1.0) { mean = xv; std = (double)xv; }                          // :15-17
      else {
        const float old_mean = mean;
        mean = old_mean + (xv - old_mean) / (float)n;                         // float32 mean (:19)
        S = S + (double)((xv - old_mean) * (xv - mean));                      // f32 product into f64 S (:20)
        std = sqrt(S / n);                                                    // :21
      }
    }
    y[(size_t)i * D + k] = (float)((double)(xv - mean) / (std + 1e-8));      // :33
  }
  __syncthreads();          // every lane has read stats[0] before lane 0 rewrites it
  if (valid) { stats[2 + k] = (double)mean; stats[2 + D + k] = S; stats[2 + 2 * D + k] = std; }
  if (k == 0) stats[0] = n;
}

__global__ void reward_scaling_kernel(const float* __restrict__ r, const uint8_t* __restrict__ done, int N,
                                      double gamma, double* __restrict__ R, double* __restrict__ stats,
                                      float* __restrict__ y) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double n = stats[0];
  float mean = (float)stats[2];
  double S = stats[3], std = stats[4];
  for (int i = 0; i < N; ++i) {
    const double rv = (double)r[i];
    R[i] = gamma * R[i] + rv;                                                 // :46
    const float xv = (float)R[i];                                             // update() casts to float32 (:13)
    n += 1.0;
    if (n == 1.0) { mean = xv; std = (double)xv; }
    else {
      const float old_mean = mean;
      mean = old_mean + (xv - old_mean) / (float)n;
      S = S + (double)((xv - old_mean) * (xv - mean));
      std = sqrt(S / n);
    }
    y[i] = (float)(rv / (std + 1e-8));                                        // :48
    if (done && done[i]) R[i] = 0.0;                                          // reset() at the next episode start
  }
  stats[0] = n; stats[2] = (double)mean; stats[3] = S; stats[4] = std;
}

// ---------------------------------------------------------------- TD3 / DDPG ---
// Gaussian action noise.  mode 0 = exploration in select_action (ddpg_pendulum.py:143-147,
// td3_pendulum.py:164-168): numpy float64 — clip(float64(mu) + eps * std, -bound, bound), stored as
// float32 when the transition is batched (:163).  mode 1 = target-policy smoothing
// (td3_pendulum.py:191-196): torch float32 — n = clamp(eps * std, -clip, clip); clamp(mu + n, +-bound).
// eps: explicit N(0,1) draws (f64, parity mode) or NULL -> Box-Muller on Philox(seed, counter, element).
__global__ __launch_bounds__(kBlock) void noisy_action_kernel(const float* __restrict__ mu, const double* __restrict__ eps,
                                                              uint64_t seed, uint64_t counter, int64_t n, int mode,
                                                              double std, float noise_clip, float bound,
                                                              float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double e = eps ? eps[i] : (double)box_muller(seed, counter, 2u, (uint32_t)i);
  if (mode == 0) {
    double a = (double)mu[i] + e * std;
    a = a < -(double)bound ? -(double)bound : (a > (double)bound ? (double)bound : a);
    out[i] = (float)a;
  } else {
    float nz = (float)e * (float)std;
    nz = fminf(fmaxf(nz, -noise_clip), noise_clip);
    out[i] = fminf(fmaxf(mu[i] + nz, -bound), bound);
  }
}

// F.mse_loss(q, y) forward + backward for one critic (ddpg_pendulum.py:178-179): dq = 2 (q - y) / B.
__global__ __launch_bounds__(kBlock) void mse_kernel(const float* __restrict__ q, const float* __restrict__ y, int B,
                                                     float* __restrict__ dq, double* __restrict__ partials, double* __restrict__ direct) {
  double acc[1] = {0.0};
  const float invB = 1.0f / (float)B;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    const float e = q[b] - y[b];
    dq[b] = 2.0f * e * invB;
    acc[0] += (double)(e * e);
  }
  block_partials<1>(acc, partials, direct);
}

// actor loss -mean(Q(s, mu(s))) (ddpg_pendulum.py:185, td3_pendulum.py:213): dq = -1/B, sum = sum q.
__global__ __launch_bounds__(kBlock) void neg_mean_kernel(const float* __restrict__ q, int B, float* __restrict__ dq,
                                                          double* __restrict__ partials, double* __restrict__ direct) {
  double acc[1] = {0.0};
  const float g = -1.0f / (float)B;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    dq[b] = g;
    acc[0] += (double)q[b];
  }
  block_partials<1>(acc, partials, direct);
}

// ------------------------------------------------------------ discrete SAC ---
// sac_cartpole.py:148-227 — expectation over the A actions instead of a reparameterised sample.  All float32
// (log_alpha is a float32 scalar there, :118-120).  A <= 8; sums over actions run in index order.
constexpr int kMaxA = 8;

// :171-181  y = r + gamma (1 - done) (sum_a p'(a) min(Q1', Q2')(a) + alpha H(p')),  log p = log(p + 1e-8)
__global__ __launch_bounds__(kBlock) void dsac_target_kernel(const float* __restrict__ probs_n, const float* __restrict__ q1n,
                                                             const float* __restrict__ q2n, const float* __restrict__ rew,
                                                             const float* __restrict__ done, const float* __restrict__ log_alpha,
                                                             int B, int A, float gamma, float* __restrict__ y) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const float alpha = det_expf(log_alpha[0]);
  float ent = 0.0f, minq = 0.0f;
  for (int k = 0; k < A; ++k) {
    const float p = probs_n[(size_t)b * A + k];
    ent += p * det_logf(p + 1e-8f);
    minq += p * fminf(q1n[(size_t)b * A + k], q2n[(size_t)b * A + k]);
  }
  const float nv = minq + alpha * (-ent);
  y[b] = rew[b] + gamma * (1.0f - done[b]) * nv;
}

// :183-186  F.mse_loss(q.gather(1, a), y) for both critics: dq[b, k] = (k == a_b) 2 (q - y) / B
__global__ __launch_bounds__(kBlock) void dsac_critic_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                             const int32_t* __restrict__ act, const float* __restrict__ y, int B,
                                                             int A, float* __restrict__ dq1, float* __restrict__ dq2,
                                                             double* __restrict__ partials, double* __restrict__ direct) {
  double acc[2] = {0.0, 0.0};
  const float invB = 1.0f / (float)B;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    const int a = act[b];
    const float e1 = q1[(size_t)b * A + a] - y[b], e2 = q2[(size_t)b * A + a] - y[b];
    for (int k = 0; k < A; ++k) {
      dq1[(size_t)b * A + k] = k == a ? 2.0f * e1 * invB : 0.0f;
      dq2[(size_t)b * A + k] = k == a ? 2.0f * e2 * invB : 0.0f;
    }
    acc[0] += (double)(e1 * e1); acc[1] += (double)(e2 * e2);
  }
  block_partials<2>(acc, partials, direct);
}

// :196-203  L = mean(-alpha H(p) - sum_a p(a) min(Q1, Q2)(a));  dL/dp_k = (alpha (log(p_k + 1e-8) + p_k/(p_k + 1e-8)) - m_k)/B
// sums: [sum (-alpha H - min_q), sum H]   (the second feeds the temperature loss :209-211)
__global__ __launch_bounds__(kBlock) void dsac_actor_kernel(const float* __restrict__ probs, const float* __restrict__ q1,
                                                            const float* __restrict__ q2, const float* __restrict__ log_alpha,
                                                            int B, int A, float* __restrict__ dprobs,
                                                            double* __restrict__ partials, double* __restrict__ direct) {
  double acc[2] = {0.0, 0.0};
  const float invB = 1.0f / (float)B;
  const float alpha = det_expf(log_alpha[0]);
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    float ent = 0.0f, minq = 0.0f;
    for (int k = 0; k < A; ++k) {
      const float p = probs[(size_t)b * A + k];
      const float lp = det_logf(p + 1e-8f);
      const float m = fminf(q1[(size_t)b * A + k], q2[(size_t)b * A + k]);
      ent += p * lp;
      minq += p * m;
      dprobs[(size_t)b * A + k] = (alpha * (lp + p / (p + 1e-8f)) - m) * invB;
    }
    ent = -ent;
    acc[0] += (double)(-alpha * ent - minq);
    acc[1] += (double)ent;
  }
  block_partials<2>(acc, partials, direct);
}

// :209-215  L_alpha = mean(exp(log_alpha) (H - H_target).detach()); Adam on the float32 scalar.
__global__ void dsac_alpha_kernel(float* __restrict__ log_alpha, float* __restrict__ m, float* __restrict__ v,
                                  const double* __restrict__ sums, int B, float target_entropy, float lr, float b1,
                                  float b2, float eps, int64_t step, const double* __restrict__ bias_dev,
                                  double* __restrict__ loss_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float alpha = det_expf(log_alpha[0]);
  const float mean_gap = (float)(sums[1] / (double)B) - target_entropy;
  if (loss_out) loss_out[0] = (double)(alpha * mean_gap);
  const float g = alpha * mean_gap;                       // d/d log_alpha of exp(log_alpha) * c
  const float mm = b1 * m[0] + (1.0f - b1) * g;
  const float vv = b2 * v[0] + (1.0f - b2) * g * g;
  m[0] = mm; v[0] = vv;
  const double bc1 = bias_dev ? bias_dev[0] : 1.0 - pow((double)b1, (double)step);
  const double bc2 = bias_dev ? bias_dev[1] : 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float denom = (float)(sqrt((double)vv) / sqrt(bc2)) + eps;
  log_alpha[0] = log_alpha[0] - step_size * (mm / denom);
}

}  // namespace

extern "C" {

int gymrl_noisy_noise(const float* eps_in_raw, const float* eps_out_raw, uint64_t seed, uint64_t counter,
                      int in_features, int out_features, float* w_eps_out, float* b_eps_out,
                      const uint64_t* counter_dev, void* stream_) {
  if (!w_eps_out || !b_eps_out || in_features <= 0 || out_features <= 0) return -22;
  if ((eps_in_raw == nullptr) != (eps_out_raw == nullptr)) return -22;
  hipLaunchKernelGGL(noisy_noise_kernel, dim3(grid_for(in_features * out_features)), dim3(kBlock), 0,
                     (hipStream_t)stream_, eps_in_raw, eps_out_raw, seed, counter, in_features, out_features,
                     w_eps_out, b_eps_out, counter_dev);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_epsilon_greedy(const float* q, const float* u, uint64_t seed, uint64_t counter, int64_t env_id0,
                         int n, int A, float epsilon, int32_t* act_out, void* stream_) {
  if (!q || !act_out || n < 0 || A <= 0) return -22;
  if (n == 0) return 0;
  hipLaunchKernelGGL(epsilon_greedy_kernel, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, q, u,
                     seed, counter, env_id0, n, A, epsilon, act_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_dqn_td_loss(const float* q, const float* q_next_online, const float* q_next_target,
                      const int32_t* act, const float* rew, const float* flag, const float* w, int B, int A,
                      double gamma_n, float* td_out, float* dq_out, double* loss_sum, void* workspace,
                      void* stream_) {
  if (!q || !q_next_target || !act || !rew || !flag || !td_out || !dq_out || B <= 0 || A <= 0 ||
      (loss_sum && !workspace))
    return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  double* parts = loss_sum ? (double*)workspace : nullptr;
  hipLaunchKernelGGL(dqn_td_kernel, dim3(nb), dim3(kBlock), 0, stream, q, q_next_online, q_next_target, act,
                     rew, flag, w, B, A, (float)gamma_n, td_out, dq_out, parts, (loss_sum && nb == 1) ? loss_sum : nullptr);
  if (loss_sum && nb > 1) hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(kBlock), 0, stream, parts, nb, loss_sum);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sac_sample_fwd(const float* mean, const float* log_std, const float* eps, int B, int A, float bound,
                         float* action_out, float* logp_out, void* stream_) {
  if (!mean || !log_std || !eps || !action_out || !logp_out || B < 0 || A <= 0) return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(sac_sample_fwd_kernel, dim3(cdiv(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, mean,
                     log_std, eps, B, A, bound, action_out, logp_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sac_sample_bwd(const float* mean, const float* log_std, const float* eps, const float* d_action,
                         const float* d_logp, int B, int A, float bound, float* d_mean_out,
                         float* d_log_std_out, void* stream_) {
  if (!mean || !log_std || !eps || !d_mean_out || !d_log_std_out || B < 0 || A <= 0) return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(sac_sample_bwd_kernel, dim3(cdiv(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, mean,
                     log_std, eps, d_action, d_logp, B, A, bound, d_mean_out, d_log_std_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sac_target(const float* rew, const float* done, const float* q1n, const float* q2n,
                     const float* logp_n, const double* log_alpha, int B, double gamma, float* y_out,
                     void* stream_) {
  if (!rew || !done || !q1n || !q2n || !logp_n || !log_alpha || !y_out || B < 0) return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(sac_target_kernel, dim3(cdiv(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, rew, done,
                     q1n, q2n, logp_n, log_alpha, B, (float)gamma, y_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sac_critic_loss(const float* q1, const float* q2, const float* y, int B, float* dq1_out,
                          float* dq2_out, double* sums, void* workspace, void* stream_) {
  if (!q1 || !q2 || !y || !dq1_out || !dq2_out || !sums || !workspace || B <= 0) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  hipLaunchKernelGGL(sac_critic_kernel, dim3(nb), dim3(kBlock), 0, stream, q1, q2, y, B, dq1_out, dq2_out,
                     (double*)workspace, nb == 1 ? sums : nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, sums);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sac_actor_loss(const float* logp, const float* q1, const float* q2, const double* log_alpha, int B,
                         double target_entropy, float* dlogp_out, float* dq1_out, float* dq2_out,
                         double* sums, void* workspace, void* stream_) {
  if (!logp || !q1 || !q2 || !log_alpha || !dlogp_out || !dq1_out || !dq2_out || !sums || !workspace || B <= 0)
    return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  hipLaunchKernelGGL(sac_actor_kernel, dim3(nb), dim3(kBlock), 0, stream, logp, q1, q2, log_alpha, B,
                     (float)target_entropy, dlogp_out, dq1_out, dq2_out, (double*)workspace, nb == 1 ? sums + 1 : nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, sums + 1);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_sac_alpha_step(double* log_alpha, double* m, double* v, const double* sums, int B, double lr,
                         double beta1, double beta2, double eps, int64_t step, const double* bias_dev,
                         double* alpha_loss_out, void* stream_) {
  if (!log_alpha || !m || !v || !sums || B <= 0 || (step < 1 && !bias_dev)) return -22;
  if (bias_dev) step = 1;
  const double bc1 = 1.0 - __builtin_pow(beta1, (double)step);
  const double bc2_sqrt = __builtin_sqrt(1.0 - __builtin_pow(beta2, (double)step));
  hipLaunchKernelGGL(sac_alpha_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, log_alpha, m, v, sums,
                     B, lr, beta1, beta2, eps, bc1, bc2_sqrt, bias_dev, alpha_loss_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_running_norm(const float* x, int N, int D, double* stats, int update, float* y_out, void* stream_) {
  if (!x || !stats || !y_out || N < 0 || D <= 0 || D > 1024) return -22;
  if (N == 0) return 0;
  hipLaunchKernelGGL(running_norm_kernel, dim3(1), dim3(((D + 63) / 64) * 64), 0, (hipStream_t)stream_, x, N, D,
                     stats, update, y_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_reward_scaling(const float* r, const uint8_t* done, int N, double gamma, double* R, double* stats,
                         float* y_out, void* stream_) {
  if (!r || !R || !stats || !y_out || N < 0) return -22;
  if (N == 0) return 0;
  hipLaunchKernelGGL(reward_scaling_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, r, done, N, gamma, R,
                     stats, y_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_noisy_action(const float* mu, const double* eps, uint64_t seed, uint64_t counter, int64_t n, int mode,
                       double std, double noise_clip, double bound, float* out, void* stream_) {
  if (!mu || !out || n < 0 || (mode != 0 && mode != 1) || std < 0.0 || bound <= 0.0 || (mode == 1 && noise_clip <= 0.0))
    return -22;
  if (n == 0) return 0;
  hipLaunchKernelGGL(noisy_action_kernel, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, mu, eps, seed,
                     counter, n, mode, std, (float)noise_clip, (float)bound, out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_mse_loss(const float* q, const float* y, int B, float* dq_out, double* sum_out, void* workspace,
                   void* stream_) {
  if (!q || !y || !dq_out || !sum_out || !workspace || B <= 0) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(kBlock), 0, stream, q, y, B, dq_out, (double*)workspace, nb == 1 ? sum_out : nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, sum_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_neg_mean_loss(const float* q, int B, float* dq_out, double* sum_out, void* workspace, void* stream_) {
  if (!q || !dq_out || !sum_out || !workspace || B <= 0) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  hipLaunchKernelGGL(neg_mean_kernel, dim3(nb), dim3(kBlock), 0, stream, q, B, dq_out, (double*)workspace, nb == 1 ? sum_out : nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(finalize_kernel<1>, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, sum_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_dsac_target(const float* probs_n, const float* q1n, const float* q2n, const float* rew, const float* done,
                      const float* log_alpha, int B, int A, double gamma, float* y_out, void* stream_) {
  if (!probs_n || !q1n || !q2n || !rew || !done || !log_alpha || !y_out || B <= 0 || A <= 0 || A > kMaxA) return -22;
  hipLaunchKernelGGL(dsac_target_kernel, dim3(cdiv(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_, probs_n, q1n, q2n,
                     rew, done, log_alpha, B, A, (float)gamma, y_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_dsac_critic_loss(const float* q1, const float* q2, const int32_t* act, const float* y, int B, int A,
                           float* dq1_out, float* dq2_out, double* sums, void* workspace, void* stream_) {
  if (!q1 || !q2 || !act || !y || !dq1_out || !dq2_out || !sums || !workspace || B <= 0 || A <= 0 || A > kMaxA) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  hipLaunchKernelGGL(dsac_critic_kernel, dim3(nb), dim3(kBlock), 0, stream, q1, q2, act, y, B, A, dq1_out, dq2_out,
                     (double*)workspace, nb == 1 ? sums : nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, sums);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_dsac_actor_loss(const float* probs, const float* q1, const float* q2, const float* log_alpha, int B, int A,
                          float* dprobs_out, double* sums, void* workspace, void* stream_) {
  if (!probs || !q1 || !q2 || !log_alpha || !dprobs_out || !sums || !workspace || B <= 0 || A <= 0 || A > kMaxA) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = grid_for(B);
  hipLaunchKernelGGL(dsac_actor_kernel, dim3(nb), dim3(kBlock), 0, stream, probs, q1, q2, log_alpha, B, A, dprobs_out,
                     (double*)workspace, nb == 1 ? sums : nullptr);
  if (nb > 1)
    hipLaunchKernelGGL(finalize_kernel<2>, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, sums);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_dsac_alpha_step(float* log_alpha, float* m, float* v, const double* sums, int B, double target_entropy,
                          double lr, double beta1, double beta2, double eps, int64_t step, const double* bias_dev,
                          double* alpha_loss_out, void* stream_) {
  if (!log_alpha || !m || !v || !sums || B <= 0 || (step <= 0 && !bias_dev)) return -22;
  hipLaunchKernelGGL(dsac_alpha_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, log_alpha, m, v, sums, B,
                     (float)target_entropy, (float)lr, (float)beta1, (float)beta2, (float)eps, step, bias_dev, alpha_loss_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
