// policy_device.hpp — row helpers, log-softmax and the categorical draw shared by ppo.hip and the
// persistent rollout kernel.
#pragma once
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

namespace gymrl {

template <int A>
__device__ __forceinline__ void load_row(const float* __restrict__ p, size_t row, float (&z)[A]) {
  if constexpr (A == 4) {
    const float4 v = reinterpret_cast<const float4*>(p)[row];
    z[0] = v.x; z[1] = v.y; z[2] = v.z; z[3] = v.w;
  } else if constexpr (A == 2) {
    const float2 v = reinterpret_cast<const float2*>(p)[row];
    z[0] = v.x; z[1] = v.y;
  } else {
#pragma unroll
    for (int k = 0; k < A; ++k) z[k] = p[row * A + k];
  }
}
template <int A>
__device__ __forceinline__ void store_row(float* __restrict__ p, size_t row, const float (&z)[A]) {
  if constexpr (A == 4) {
    reinterpret_cast<float4*>(p)[row] = make_float4(z[0], z[1], z[2], z[3]);
  } else if constexpr (A == 2) {
    reinterpret_cast<float2*>(p)[row] = make_float2(z[0], z[1]);
  } else {
#pragma unroll
    for (int k = 0; k < A; ++k) p[row * A + k] = z[k];
  }
}

// log-softmax pieces shared by all kernels: ln_k = z_k - lse, p_k = e_k / s.
template <int A>
__device__ __forceinline__ void log_softmax(const float (&z)[A], float (&ln)[A], float (&p)[A],
                                            float& H) {
  float m = z[0];
#pragma unroll
  for (int k = 1; k < A; ++k) m = fmaxf(m, z[k]);
  float e[A];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < A; ++k) { e[k] = det_expf(z[k] - m); s += e[k]; }
  const float lse = m + det_logf(s);
  H = 0.0f;
#pragma unroll
  for (int k = 0; k < A; ++k) {
    ln[k] = z[k] - lse;
    p[k] = e[k] / s;
    H -= p[k] * ln[k];
  }
}

// L1 (ppo_lunarlander.py:278-300, :309-322) for ONE sample: clipped surrogate with dual clip, value and entropy
// terms, their gradient w.r.t. the logits row and the value, and the five metric terms.  Shared by ppo_loss_kernel
// and the fused heads + loss + backward pass, so both produce the same bits.  `ad` is the (already normalised)
// advantage, invB = 1 / minibatch size.
template <int A>
__device__ __forceinline__ void ppo_loss_row(const float (&z)[A], float v, int a, float lpo, float ad, float rt,
                                             float invB, const gymrl_ppo_cfg& cfg, float (&dz)[A], float& dv,
                                             float& m_obj, float& m_val, float& m_ent, float& m_clip, float& m_kl) {
  float ln[A], p[A], H;
  log_softmax<A>(z, ln, p, H);
  float lp = ln[0];
#pragma unroll
  for (int k = 1; k < A; ++k) if (a == k) lp = ln[k];
  const float lo = 1.0f - cfg.clip_eps, hi = 1.0f + cfg.clip_eps;
  const float ratio = det_expf(lp - lpo);
  const float s1 = ratio * ad;
  const float rc = fminf(fmaxf(ratio, lo), hi);
  const float s2 = rc * ad;
  const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;   // clamp passes grad on [lo, hi]
  const float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);     // torch.min tie: 1/2, 1/2
  const float ms = fminf(s1, s2);
  float dms_dr = w1 * ad + (1.0f - w1) * ad * inr;
  float obj = ms;
  if (ad < 0.0f) {
    const float dc = cfg.dual_clip * ad;
    obj = fmaxf(ms, dc);
    const float wm = ms > dc ? 1.0f : (ms == dc ? 0.5f : 0.0f);   // torch.max tie
    dms_dr *= wm;
  }
  // dL/dlp = -(1/B) * dobj/dr * r
  const float g_lp = -invB * dms_dr * ratio;
  const float g_H = -cfg.entropy_coef * invB;
#pragma unroll
  for (int k = 0; k < A; ++k) {
    const float onehot = (a == k) ? 1.0f : 0.0f;
    dz[k] = g_lp * (onehot - p[k]) + g_H * (-p[k] * (ln[k] + H));
  }
  const float dvr = v - rt;
  dv = cfg.value_coef * 2.0f * dvr * invB;
  m_obj = obj;
  m_val = cfg.value_coef * (dvr * dvr);
  m_ent = H;
  m_clip = (ratio < lo || ratio > hi) ? 1.0f : 0.0f;
  m_kl = lpo - lp;
}

// P2 (ppo_lunarlander.py:92-104): Categorical(logits) -> (action, log_prob, entropy).  sample =
// argmax_k p_k / q_k with q ~ Exp(1): `noise_row` (explicit draws, parity mode) or Philox keyed by
// (seed, global env id, counter).  Shared by categorical_sample_kernel and the persistent rollout.
template <int A>
__device__ __forceinline__ int categorical_pick(const float (&z)[A], const float* __restrict__ noise_row, uint64_t seed,
                                                uint64_t env, uint64_t counter, int deterministic, float& lp,
                                                float& H) {
  float ln[A], p[A];
  log_softmax<A>(z, ln, p, H);
  int a = 0;
  if (deterministic) {
    float best = z[0];
#pragma unroll
    for (int k = 1; k < A; ++k) if (z[k] > best) { best = z[k]; a = k; }
  } else {
    float q[A];
    if (noise_row) {
      load_row<A>(noise_row, 0, q);
    } else {
#pragma unroll
      for (int blk = 0; blk < (A + 3) / 4; ++blk) {
        const u32x4 r = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), (uint32_t)counter,
                                   RNG_POLICY | ((uint32_t)((counter >> 32) & 0x3FFFFFu) << 2) | (uint32_t)blk);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (blk * 4 + k < A) q[blk * 4 + k] = -det_logf(u01f_open0(w[k]));
      }
    }
    float best = p[0] / q[0];
#pragma unroll
    for (int k = 1; k < A; ++k) {
      const float c = p[k] / q[k];
      if (c > best) { best = c; a = k; }
    }
  }
  lp = ln[0];
#pragma unroll
  for (int k = 1; k < A; ++k) if (a == k) lp = ln[k];
  return a;
}

}  // namespace gymrl
