// ppo.hip — categorical policy head + fused PPO loss forward/backward (gfx950).
//
//   P2  gymrl_categorical_sample       ppo_lunarlander.py:92-104
//   L1  gymrl_ppo_loss_fwd_bwd         ppo_lunarlander.py:110-117, 278-300, 309-322
//   L3  gymrl_ppo_full_loss_fwd_bwd    ppo_full_lunarlander.py:575-652
//
// One lane = one sample; the A (<= 8) logits of a sample live in registers, the
// row is fetched as one 16-B load when A == 4 (1 KiB per wave-instruction).
// Everything here is HBM-bound: 56 algorithmic bytes per sample for L1
// (logits 16 + v 4 + act 4 + logp_old 4 + adv 4 + ret 4 read, dlogits 16 + dv 4
// written); metrics go wave-shuffle -> LDS -> one f64 atomic per block and term.
// exp/log are the bit-reproducible det_* forms so that the CPU oracle can
// reproduce integer action draws exactly.
#include "gymrl_device.hpp"
#include "policy_device.hpp"
#include "../../include/gymrl.h"

using namespace gymrl;

namespace {

constexpr int kBlock = 256;

// K per-thread doubles -> one row of K doubles per block in partials[block][K].
template <int K, int BLOCK>
__device__ __forceinline__ void block_partials(double (&v)[K], double* __restrict__ partials) {
  __shared__ double sm[K][BLOCK / 64];
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
    for (int w = 0; w < BLOCK / 64; ++w) s += sm[threadIdx.x][w];
    partials[(size_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------- P2 sample ----
struct OnlineGae {   // device-side view of gymrl_gae_online (rew_prev == nullptr: disabled)
  const float* rew_prev; const uint8_t* done_prev; const float* val_prev;
  double* running; double2* agg_row; double gamma, gl; int first, last;
  double* running2; double2* agg_row2; double gl2;      // decoupled-lambda mode (G3): second map, nullptr = off
};

template <int A>
__global__ __launch_bounds__(kBlock) void categorical_sample_kernel(
    const float* __restrict__ logits, const float* __restrict__ value_in,
    const float* __restrict__ noise_exp, uint64_t seed, uint64_t counter, int64_t env_id0, int n,
    int deterministic, int32_t* __restrict__ act_out, float* __restrict__ logp_out,
    float* __restrict__ ent_out, float* __restrict__ value_out, OnlineGae og) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  // fused producer side of GAE: value_in is V_t, which completes step t-1's delta
  if (og.rew_prev)
  {
    gae_online_compose(og.rew_prev[i], og.done_prev[i], og.val_prev[i], value_in[i], og.gamma, og.gl, og.first,
                       og.last, og.running, og.agg_row, n, i);
    if (og.running2)
      gae_online_compose(og.rew_prev[i], og.done_prev[i], og.val_prev[i], value_in[i], og.gamma, og.gl2, og.first,
                         og.last, og.running2, og.agg_row2, n, i);
  }
  float z[A], H, lp;
  load_row<A>(logits, i, z);
  const int a = categorical_pick<A>(z, noise_exp ? noise_exp + (size_t)i * A : nullptr, seed, (uint64_t)(env_id0 + i),
                                    counter, deterministic, lp, H);
  act_out[i] = a;
  logp_out[i] = lp;
  if (ent_out) ent_out[i] = H;
  if (value_out && value_in) value_out[i] = value_in[i];
}

// ---------------------------------------------------------------- L1 loss ---
template <int A>
__global__ __launch_bounds__(kBlock) void ppo_loss_kernel(
    const float* __restrict__ logits, const float* __restrict__ value,
    const int32_t* __restrict__ idx, const int32_t* __restrict__ act,
    const float* __restrict__ logp_old, const float* __restrict__ adv,
    const float* __restrict__ ret, const double* __restrict__ adv_moments, int B,
    gymrl_ppo_cfg cfg, float* __restrict__ dlogits_out, float* __restrict__ dvalue_out,
    double* __restrict__ partials) {
  double met[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  // whole-rollout advantage normalisation constants (ppo_lunarlander.py:236): one f64
  // sqrt/divide per workgroup instead of per sample.
  __shared__ double s_norm[2];
  if (adv_moments && threadIdx.x == 0) {
    const double cnt = adv_moments[0];
    const double mean = adv_moments[1] / cnt;
    double var = adv_moments[2] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    s_norm[0] = mean; s_norm[1] = sqrt(var) + 1e-8;
  }
  // (the barrier publishing s_norm sits AFTER the first sample's loads are issued, so the two
  //  memory latencies overlap; at one sample per lane this is most of the kernel's critical path)
  bool synced = false;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B || !synced; b += gridDim.x * kBlock) {
    const bool live = b < B;
    const int bb = live ? b : 0;
    float z[A];
    load_row<A>(logits, bb, z);
    const float v = value[bb];
    const int i = idx ? idx[bb] : bb;
    const int a = act[i];
    const float lpo = logp_old[i];
    float ad = adv[i];
    const float rt = ret[i];
    if (!synced) { __syncthreads(); synced = true; }
    if (!live) break;
    if (adv_moments) ad = (float)(((double)ad - s_norm[0]) / s_norm[1]);
    const float invB = 1.0f / (float)B;
    float dz[A], dvo, m_obj, m_val, m_ent, m_clip, m_kl;
    ppo_loss_row<A>(z, v, a, lpo, ad, rt, invB, cfg, dz, dvo, m_obj, m_val, m_ent, m_clip, m_kl);
    store_row<A>(dlogits_out, b, dz);
    dvalue_out[b] = dvo;
    met[0] += -(double)m_obj;
    met[1] += (double)m_val;
    met[2] += (double)m_ent;
    met[3] += (double)m_clip;
    met[4] += (double)m_kl;
  }
  if (partials) block_partials<5, kBlock>(met, partials);
}

// ---------------------------------------------------------------- L3 loss ---
template <int A>
__global__ __launch_bounds__(kBlock) void ppo_full_loss_kernel(
    const float* __restrict__ logits, const float* __restrict__ value,
    const int32_t* __restrict__ idx, const int32_t* __restrict__ act,
    const float* __restrict__ logp_old, const float* __restrict__ ent_old,
    const float* __restrict__ adv, const float* __restrict__ ret, int B, gymrl_ppo_full_cfg cfg,
    const float* __restrict__ corr_mul, float* __restrict__ dlogits_out, float* __restrict__ dvalue_out,
    double* __restrict__ partials) {
  double met[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    float z[A], ln[A], p[A], H;
    load_row<A>(logits, b, z);
    const float v = value[b];
    const int i = idx ? idx[b] : b;
    const int a = act[i];
    const float lpo = logp_old[i];
    const float eo = ent_old[i];
    const float ad = adv[i];
    const float rt = ret[i];
    log_softmax<A>(z, ln, p, H);
    float lp = ln[0];
#pragma unroll
    for (int k = 1; k < A; ++k) if (a == k) lp = ln[k];

    const float invB = 1.0f / (float)B;
    // entropy-ratio mask (no gradient) — :586-590
    const float er = H / (eo + 1e-8f);
    const float erc = (er > (1.0f - cfg.erc_beta_low) && er < (1.0f + cfg.erc_beta_high)) ? 1.0f : 0.0f;
    float corr = erc;
    if (corr_mul) corr *= corr_mul[b];               // covariance clip (:611-616): rows picked on the host side of the API
    const float ratio = det_expf(lp - lpo);
    const float lo = 1.0f - cfg.clip_eps_min, hi = 1.0f + cfg.clip_eps_max;
    const float r1 = fminf(fmaxf(ratio, 0.0f), cfg.dual_clip);
    const float r2 = fminf(fmaxf(ratio, lo), hi);
    const float s1 = r1 * ad, s2 = r2 * ad;
    const float in1 = (ratio >= 0.0f && ratio <= cfg.dual_clip) ? 1.0f : 0.0f;
    const float in2 = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    const float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
    const float ms = fminf(s1, s2);
    const float dms_dr = w1 * ad * in1 + (1.0f - w1) * ad * in2;
    const float g_lp = -invB * corr * dms_dr * ratio;
    const float g_H = -(cfg.entropy_coef_dev ? cfg.entropy_coef_dev[0] : cfg.entropy_coef) * invB * corr;
    float dz[A];
#pragma unroll
    for (int k = 0; k < A; ++k) {
      const float onehot = (a == k) ? 1.0f : 0.0f;
      dz[k] = g_lp * (onehot - p[k]) + g_H * (-p[k] * (ln[k] + H));
    }
    store_row<A>(dlogits_out, b, dz);
    const float dvr = v - rt;
    dvalue_out[b] = corr * dvr * invB;

    met[0] += (double)(-ms * corr);
    met[1] += (double)(0.5f * corr * (dvr * dvr));
    met[2] += (double)(H * corr);
    met[3] += (ratio < lo || ratio > hi) ? (double)corr : 0.0;
    met[4] += (double)(lpo - lp);
    met[5] += 1.0 - (double)erc;                     // erc_clip_frac is over the entropy-ratio mask alone (:652)
    met[6] += (double)lp;
    met[7] += (double)ad;
    met[8] += (double)lp * (double)ad;
  }
  if (partials) block_partials<9, kBlock>(met, partials);
}

// ------------------------------------------------- recurrent-PPO (L4) loss ---
// ppo_lstm_lunarlander.py:716-776: the L3 terms, but every mean is a masked mean over the
// entropy-ratio mask (sum / count, 0 when the mask is empty :646-655) and the value loss is the
// clipped one (:763-770).  The count is an integer, so pass 1 adds it up with atomics
// (order-independent); pass 2 reads it.
template <int A>
__global__ __launch_bounds__(kBlock) void erc_count_kernel(const float* __restrict__ logits,
                                                           const int32_t* __restrict__ idx,
                                                           const float* __restrict__ ent_old, int B,
                                                           gymrl_ppo_full_cfg cfg, const float* __restrict__ corr_mul,
                                                           uint32_t* __restrict__ count) {
  uint32_t c = 0;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    float z[A], ln[A], p[A], H;
    load_row<A>(logits, b, z);
    log_softmax<A>(z, ln, p, H);
    const float er = H / (ent_old[idx ? idx[b] : b] + 1e-8f);
    const bool in = er > (1.0f - cfg.erc_beta_low) && er < (1.0f + cfg.erc_beta_high);
    c += (in && (!corr_mul || corr_mul[b] != 0.0f)) ? 1u : 0u;
  }
  const uint64_t m = __ballot(c & 1u);               // at most a few rows per thread: add them bit by bit
  uint32_t w = (uint32_t)__popcll(m);
  for (uint32_t bit = 1; bit < 16; ++bit) w += (uint32_t)__popcll(__ballot((c >> bit) & 1u)) << bit;
  if ((threadIdx.x & 63) == 0 && w) atomicAdd(count, w);
}

template <int A>
__global__ __launch_bounds__(kBlock) void ppo_rnn_loss_kernel(
    const float* __restrict__ logits, const float* __restrict__ value,
    const int32_t* __restrict__ idx, const int32_t* __restrict__ act,
    const float* __restrict__ logp_old, const float* __restrict__ ent_old,
    const float* __restrict__ val_old, const float* __restrict__ adv, const float* __restrict__ ret,
    int B, gymrl_ppo_full_cfg cfg, const uint32_t* __restrict__ count, const float* __restrict__ corr_mul,
    float* __restrict__ dlogits_out, float* __restrict__ dvalue_out, double* __restrict__ partials) {
  double met[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t cnt = *count;
  const float inv = cnt ? 1.0f / (float)cnt : 0.0f;
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    float z[A], ln[A], p[A], H;
    load_row<A>(logits, b, z);
    const float v = value[b];
    const int i = idx ? idx[b] : b;
    const int a = act[i];
    const float lpo = logp_old[i];
    const float eo = ent_old[i];
    const float vo = val_old[i];
    const float ad = adv[i];
    const float rt = ret[i];
    log_softmax<A>(z, ln, p, H);
    float lp = ln[0];
#pragma unroll
    for (int k = 1; k < A; ++k) if (a == k) lp = ln[k];

    const float er = H / (eo + 1e-8f);
    const float erc = (er > (1.0f - cfg.erc_beta_low) && er < (1.0f + cfg.erc_beta_high)) ? 1.0f : 0.0f;
    float corr = erc;
    if (corr_mul) corr *= corr_mul[b];               // covariance clip (:611-616): rows picked on the host side of the API
    const float ratio = det_expf(lp - lpo);
    const float lo = 1.0f - cfg.clip_eps_min, hi = 1.0f + cfg.clip_eps_max;
    const float r1 = fminf(fmaxf(ratio, 0.0f), cfg.dual_clip);
    const float r2 = fminf(fmaxf(ratio, lo), hi);
    const float s1 = r1 * ad, s2 = r2 * ad;
    const float in1 = (ratio >= 0.0f && ratio <= cfg.dual_clip) ? 1.0f : 0.0f;
    const float in2 = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    const float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
    const float ms = fminf(s1, s2);
    const float dms_dr = w1 * ad * in1 + (1.0f - w1) * ad * in2;
    const float scale = corr * inv;
    const float g_lp = -scale * dms_dr * ratio;
    const float g_H = -(cfg.entropy_coef_dev ? cfg.entropy_coef_dev[0] : cfg.entropy_coef) * scale;
    float dz[A];
#pragma unroll
    for (int k = 0; k < A; ++k) {
      const float onehot = (a == k) ? 1.0f : 0.0f;
      dz[k] = g_lp * (onehot - p[k]) + g_H * (-p[k] * (ln[k] + H));
    }
    store_row<A>(dlogits_out, b, dz);
    // clipped value loss: 0.5 * max((v - ret)^2, (v_old + clamp(v - v_old, -eps_min, eps_max) - ret)^2)
    const float dv = v - vo;
    const float vc = vo + fminf(fmaxf(dv, -cfg.clip_eps_min), cfg.clip_eps_max);
    const float inv_ = (dv >= -cfg.clip_eps_min && dv <= cfg.clip_eps_max) ? 1.0f : 0.0f;
    const float e1 = v - rt, e2 = vc - rt;
    const float l1 = e1 * e1, l2 = e2 * e2;
    const float wv = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);       // torch.max tie: 1/2, 1/2
    dvalue_out[b] = 0.5f * scale * (wv * 2.0f * e1 + (1.0f - wv) * 2.0f * e2 * inv_);

    met[0] += (double)(-ms * corr);
    met[1] += (double)(0.5f * corr * fmaxf(l1, l2));
    met[2] += (double)(H * corr);
    met[3] += (ratio < lo || ratio > hi) ? (double)corr : 0.0;
    met[4] += (double)(lpo - lp);
    met[5] += 1.0 - (double)erc;                     // erc_clip_frac is over the entropy-ratio mask alone (:652)
    met[6] += (double)lp;
    met[7] += (double)ad;
    met[8] += (double)lp * (double)ad;
    met[9] += (double)corr;
  }
  if (partials) block_partials<10, kBlock>(met, partials);
}

// metrics_sum[k] += sum over blocks of partials[block][k], fixed order (no atomics:
// 32k same-address f64 atomics cost ~400 us at B = 8.4M, more than the kernel itself).
template <int K>
__global__ __launch_bounds__(kBlock) void metrics_finalize_kernel(const double* __restrict__ partials,
                                                                  int nblocks,
                                                                  double* __restrict__ metrics_sum) {
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
    metrics_sum[threadIdx.x] += s;
  }
}

constexpr int kMaxLossBlocks = 1024;

// ------------------------------------------------------- minibatch staging ---
// P6/P7 (ppo_lunarlander.py:238-272).  A uniformly random minibatch touches one
// random row per sample; gathering five SoA arrays costs five 64-B lines per sample
// for 48 useful bytes.  So the rollout is packed ONCE into 64-B records
//   [ obs(<=12 f32) | act bits | logp_old | adv | ret ]
// (one cache line per transition, streaming pass), and each minibatch is ONE random
// line per sample gathered into contiguous SoA rows that the GEMMs and the loss
// kernel then stream.
constexpr int kRecF = 16;   // floats per packed record (64 B)

// P6 epoch shuffle (ppo_lunarlander.py:262 np.random.shuffle): perm[i] = keyed bijection of i.
__global__ __launch_bounds__(kBlock) void permutation_kernel(uint64_t seed, uint64_t counter, uint32_t M, int a, int b,
                                                             int32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < M) out[i] = (int32_t)keyed_permute(i, M, a, b, seed, counter);
}

__global__ __launch_bounds__(kBlock) void pack_rollout_kernel(
    const float* __restrict__ obs, const int32_t* __restrict__ act, const float* __restrict__ logp,
    const float* __restrict__ adv, const float* __restrict__ ret, int64_t M, int D,
    float* __restrict__ packed) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += stride) {
    float r[kRecF];
#pragma unroll
    for (int k = 0; k < 12; ++k) r[k] = k < D ? obs[i * D + k] : 0.0f;
    r[12] = __int_as_float(act[i]); r[13] = logp[i]; r[14] = adv[i]; r[15] = ret[i];
    float4* dst = reinterpret_cast<float4*>(packed + i * kRecF);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
  }
}

__global__ __launch_bounds__(kBlock) void gather_minibatch_kernel(
    const float* __restrict__ packed, const int32_t* __restrict__ idx, int B, int D,
    float* __restrict__ obs_out, int32_t* __restrict__ act_out, float* __restrict__ logp_out,
    float* __restrict__ adv_out, float* __restrict__ ret_out) {
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < B; b += gridDim.x * kBlock) {
    const float4* src = reinterpret_cast<const float4*>(packed + (size_t)idx[b] * kRecF);
    const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    const float r[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
    if (D == 8) {
      reinterpret_cast<float4*>(obs_out)[2 * (size_t)b] = q0;
      reinterpret_cast<float4*>(obs_out)[2 * (size_t)b + 1] = q1;
    } else if (D == 4) {
      reinterpret_cast<float4*>(obs_out)[b] = q0;
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k < D) obs_out[(size_t)b * D + k] = r[k];
    }
    act_out[b] = __float_as_int(q3.x); logp_out[b] = q3.y; adv_out[b] = q3.z; ret_out[b] = q3.w;
  }
}

// out[b, :] = src[idx[b], :] for rows of `quads` float4s (a minibatch of observations by index: states[mb_idx]); one lane
// per float4, so a row is `quads` adjacent lanes' 16-byte loads
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                                                             int quads, float4* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n * quads; e += (int64_t)gridDim.x * kBlock) {
    const int64_t b = e / quads;
    const int q = (int)(e - b * quads);
    out[e] = src[(int64_t)idx[b] * quads + q];
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

#define DISPATCH_A(A_, CALL)              \
  switch (A_) {                           \
    case 2: { constexpr int A = 2; CALL; } break; \
    case 3: { constexpr int A = 3; CALL; } break; \
    case 4: { constexpr int A = 4; CALL; } break; \
    case 5: { constexpr int A = 5; CALL; } break; \
    case 6: { constexpr int A = 6; CALL; } break; \
    case 7: { constexpr int A = 7; CALL; } break; \
    case 8: { constexpr int A = 8; CALL; } break; \
    default: return -22;                  \
  }

extern "C" {

int gymrl_categorical_sample(const float* logits, const float* value_in, const float* noise_exp,
                             uint64_t seed, uint64_t counter, int64_t env_id0, int n,
                             int n_actions, int deterministic, int32_t* act_out, float* logp_out,
                             float* ent_out, float* value_out, const gymrl_gae_online* online,
                             void* stream_) {
  if (!logits || !act_out || !logp_out || n < 0) return -22;
  if (n == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  OnlineGae og{};
  if (online) {
    if (!value_in || !online->rew_prev || !online->done_prev || !online->val_prev || !online->running ||
        !online->gae_workspace || online->t_prev < 0 || online->t_prev >= online->T)
      return -22;
    const int tc = gymrl_gae_chunk();
    og.rew_prev = online->rew_prev; og.done_prev = online->done_prev; og.val_prev = online->val_prev;
    og.running = online->running;
    og.agg_row = (double2*)online->gae_workspace + (size_t)(online->t_prev / tc) * n;
    og.gamma = online->gamma; og.gl = (double)(float)(online->gamma * online->lam);
    if (online->lam2 > 0.0 && online->running2) {      // G3: float64 decay factors, critic maps after the actor's
      const size_t C = (size_t)((online->T + tc - 1) / tc);
      og.gl = online->gamma * online->lam;
      og.gl2 = online->gamma * online->lam2;
      og.running2 = online->running2;
      og.agg_row2 = og.agg_row + C * (size_t)n;
    }
    og.first = (online->t_prev % tc) == 0;
    og.last = (online->t_prev % tc) == tc - 1 || online->t_prev == online->T - 1;
  }
  DISPATCH_A(n_actions,
             hipLaunchKernelGGL(categorical_sample_kernel<A>, dim3(cdiv(n, kBlock)), dim3(kBlock),
                                0, stream, logits, value_in, noise_exp, seed, counter, env_id0, n,
                                deterministic, act_out, logp_out, ent_out, value_out, og));
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_ppo_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                           const int32_t* act, const float* logp_old, const float* adv,
                           const float* ret, const double* adv_moments, int B, int n_actions,
                           const gymrl_ppo_cfg* cfg_host, float* dlogits_out, float* dvalue_out,
                           double* metrics_sum, void* workspace, void* stream_) {
  if (!logits || !value || !act || !logp_old || !adv || !ret || !cfg_host || !dlogits_out ||
      !dvalue_out || B < 0 || (metrics_sum && !workspace))
    return -22;
  if (B == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const gymrl_ppo_cfg cfg = *cfg_host;
  const int nb = cdiv(B, kBlock) < kMaxLossBlocks ? cdiv(B, kBlock) : kMaxLossBlocks;
  double* parts = (double*)workspace;   // block partials [nb][5]; with metrics_sum == NULL they are the output
  DISPATCH_A(n_actions,
             hipLaunchKernelGGL(ppo_loss_kernel<A>, dim3(nb), dim3(kBlock), 0, stream,
                                logits, value, idx, act, logp_old, adv, ret, adv_moments, B, cfg,
                                dlogits_out, dvalue_out, parts));
  if (metrics_sum)
    hipLaunchKernelGGL(metrics_finalize_kernel<5>, dim3(1), dim3(kBlock), 0, stream, parts, nb,
                       metrics_sum);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_ppo_full_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                                const int32_t* act, const float* logp_old, const float* ent_old,
                                const float* adv, const float* ret, int B, int n_actions,
                                const gymrl_ppo_full_cfg* cfg_host, const float* corr_mul, float* dlogits_out,
                                float* dvalue_out, double* metrics_sum, void* workspace,
                                void* stream_) {
  if (!logits || !value || !act || !logp_old || !ent_old || !adv || !ret || !cfg_host ||
      !dlogits_out || !dvalue_out || B < 0 || (metrics_sum && !workspace))
    return -22;
  if (B == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const gymrl_ppo_full_cfg cfg = *cfg_host;
  const int nb = cdiv(B, kBlock) < kMaxLossBlocks ? cdiv(B, kBlock) : kMaxLossBlocks;
  double* parts = (double*)workspace;
  DISPATCH_A(n_actions,
             hipLaunchKernelGGL(ppo_full_loss_kernel<A>, dim3(nb), dim3(kBlock), 0,
                                stream, logits, value, idx, act, logp_old, ent_old, adv, ret, B, cfg, corr_mul,
                                dlogits_out, dvalue_out, parts));
  if (metrics_sum)
    hipLaunchKernelGGL(metrics_finalize_kernel<9>, dim3(1), dim3(kBlock), 0, stream, parts, nb,
                       metrics_sum);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_permutation(uint64_t seed, uint64_t counter, int64_t M, int32_t* perm_out, void* stream_) {
  if (!perm_out || M < 0 || M > (int64_t)1 << 30) return -22;
  if (M == 0) return 0;
  int bits = 2;
  while (((int64_t)1 << bits) < M) ++bits;
  const int a = bits / 2, b = bits - a;
  hipLaunchKernelGGL(permutation_kernel, dim3((unsigned)((M + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream_, seed, counter, (uint32_t)M, a, b, perm_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_ppo_rnn_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                               const int32_t* act, const float* logp_old, const float* ent_old,
                               const float* val_old, const float* adv, const float* ret, int B, int n_actions,
                               const gymrl_ppo_full_cfg* cfg_host, const float* corr_mul, float* dlogits_out,
                               float* dvalue_out, double* metrics_sum, void* workspace, void* stream_) {
  if (!logits || !value || !act || !logp_old || !ent_old || !val_old || !adv || !ret || !cfg_host ||
      !dlogits_out || !dvalue_out || !workspace || B < 0 || B > (1 << 24))
    return -22;
  if (B == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const gymrl_ppo_full_cfg cfg = *cfg_host;
  const int nb = cdiv(B, kBlock) < kMaxLossBlocks ? cdiv(B, kBlock) : kMaxLossBlocks;
  double* parts = (double*)workspace;
  uint32_t* count = reinterpret_cast<uint32_t*>(parts + kMaxLossBlocks * 10);
  if (hipMemsetAsync(count, 0, sizeof(uint32_t), stream) != hipSuccess) return -1000 - (int)hipGetLastError();
  DISPATCH_A(n_actions,
             hipLaunchKernelGGL(erc_count_kernel<A>, dim3(nb), dim3(kBlock), 0, stream, logits, idx, ent_old, B,
                                cfg, corr_mul, count));
  DISPATCH_A(n_actions,
             hipLaunchKernelGGL(ppo_rnn_loss_kernel<A>, dim3(nb), dim3(kBlock), 0, stream, logits, value, idx,
                                act, logp_old, ent_old, val_old, adv, ret, B, cfg, count, corr_mul, dlogits_out,
                                dvalue_out, metrics_sum ? parts : nullptr));
  if (metrics_sum)
    hipLaunchKernelGGL(metrics_finalize_kernel<10>, dim3(1), dim3(kBlock), 0, stream, parts, nb,
                       metrics_sum);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

// out[r][k] = sum_b partials[r][b][k]: one launch reduces the block partials of a whole
// update's minibatches (instead of one tiny finalize launch per minibatch).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const double* __restrict__ partials,
                                                          int blocks_per_row, int K,
                                                          double* __restrict__ out) {
  __shared__ double sm[256];
  const double* src = partials + (size_t)blockIdx.x * blocks_per_row * K;
  for (int k = 0; k < K; ++k) {
    double a = 0.0;
    for (int b = threadIdx.x; b < blocks_per_row; b += 256) a += src[(size_t)b * K + k];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if (threadIdx.x < s2) sm[threadIdx.x] += sm[threadIdx.x + s2];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * K + k] = sm[0];
    __syncthreads();
  }
}

int gymrl_loss_blocks(int B) { return cdiv(B, kBlock) < kMaxLossBlocks ? cdiv(B, kBlock) : kMaxLossBlocks; }

int gymrl_reduce_rows(const double* partials, int rows, int blocks_per_row, int K, double* out,
                      void* stream_) {
  if (!partials || !out || rows < 0 || blocks_per_row <= 0 || K <= 0) return -22;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream_, partials,
                     blocks_per_row, K, out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_pack_rollout(const float* obs, const int32_t* act, const float* logp, const float* adv,
                       const float* ret, int64_t M, int obs_dim, float* packed, void* stream_) {
  if (!obs || !act || !logp || !adv || !ret || !packed || M < 0 || obs_dim < 1 || obs_dim > 12) return -22;
  if ((reinterpret_cast<uintptr_t>(packed) & 15) != 0) return -22;
  if (M == 0) return 0;
  int64_t nb = (M + kBlock - 1) / kBlock;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_rollout_kernel, dim3((int)nb), dim3(kBlock), 0, (hipStream_t)stream_, obs, act,
                     logp, adv, ret, M, obs_dim, packed);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_gather_minibatch(const float* packed, const int32_t* idx, int B, int obs_dim, float* obs_out,
                           int32_t* act_out, float* logp_out, float* adv_out, float* ret_out,
                           void* stream_) {
  if (!packed || !idx || !obs_out || !act_out || !logp_out || !adv_out || !ret_out || B < 0 ||
      obs_dim < 1 || obs_dim > 12)
    return -22;
  if ((reinterpret_cast<uintptr_t>(packed) & 15) != 0 || (reinterpret_cast<uintptr_t>(obs_out) & 15) != 0) return -22;
  if (B == 0) return 0;
  const int nb = cdiv(B, kBlock) < 4096 ? cdiv(B, kBlock) : 4096;
  hipLaunchKernelGGL(gather_minibatch_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream_, packed, idx,
                     B, obs_dim, obs_out, act_out, logp_out, adv_out, ret_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_gather_rows(const float* src, const int32_t* idx, int B, int row_floats, float* out, void* stream_) {
  if (!src || !idx || !out || B < 0 || row_floats < 4 || row_floats % 4 || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15))
    return -22;
  if (B == 0) return 0;
  const int quads = row_floats / 4;
  const int64_t want = ((int64_t)B * quads + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(kBlock), 0, (hipStream_t)stream_,
                     reinterpret_cast<const float4*>(src), idx, (int64_t)B, quads, reinterpret_cast<float4*>(out));
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
