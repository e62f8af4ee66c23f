/*
 * gymrl_oracle.c — CPU restatement of the gymRL hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for libgymrl_hip.so: plain scalar C, one element at a
 * time, in the reference's operation order.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (gymrl_amd/) never does.
 *
 * Pinning: every learner-side function below is checked against golden vectors
 * produced by importing the reference's own Python (tests/golden/make_golden.py,
 * tests/test_oracle_golden.py).  The env functions (CartPole / Pendulum /
 * LunarLander) restate gymnasium's published dynamics; gymnasium and Box2D are
 * third-party, absent from /root/reference and not installable here, so for
 * the env rows PARITY IS UNPINNED.
 *
 * Each function cites the reference file:line (under the gymRL tree) it follows.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() is the only fused op).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================ math ==== */
/* Reproducible f32 exp/log/sincos/tanh: the SPEC both sides implement so that
 * integer action draws can be compared bit-for-bit (torch itself is only matched
 * to 1e-5; see tests/test_oracle_golden.py).  Cephes single-precision kernels. */
float orc_expf(float x) {
  if (!(x > -87.33654f)) return (x != x) ? x : 0.0f;
  if (x > 88.72283f) return INFINITY;
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, r * r, r) + 1.0f;
  return ldexpf(y, (int)n);
}

float orc_logf(float x) {
  int e;
  float m = frexpf(x, &e);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = fmaf(p, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = p * m * z;
  float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(-0.5f, z, y);
  float r = m + y;
  return fmaf(fe, 0.693359375f, r);
}

void orc_sincosf(float x, float* s, float* c) {
  float q = rintf(x * 0.636619772367581343f);
  int qi = (int)q;
  float r = fmaf(q, -1.5703125f, x);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188e-8f, r);
  float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float sn = fmaf(ps * z, r, r);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float cs = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  float ss = (qi & 1) ? cs : sn;
  float cc = (qi & 1) ? sn : cs;
  if (qi & 2) ss = -ss;
  if ((qi + 1) & 2) cc = -cc;
  *s = ss; *c = cc;
}

/* float64 sin/cos of the classic-control envs: the operation-by-operation twin of det_sincos in
 * gymrl_amd/csrc/gymrl_device.hpp (fdlibm pio2_1 / pio2_1t reduction and kernel coefficients, fma Horner chains).
 * fma() is correctly rounded in glibc whether or not the host has the instruction, so the result does not depend on
 * the box.  <= 1.6 ulp from sinl / cosl on [-100, 100]: the published CartPole / Pendulum equations evaluated with
 * libm (tests/golden/classic_micro.npz) are still met to 1e-6 and better. */
void orc_sincos(double x, double* s, double* c) {
  double q = rint(x * 0.63661977236758134308);
  long long qi = (long long)q;
  double r = fma(q, -1.57079632673412561417e+00, x);
  r = fma(q, -6.07710050650619224932e-11, r);
  double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  double sn = fma(ps * z, r, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma(pc, z, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  double cs = fma(pc * z, z, fma(-0.5, z, 1.0));
  double ss = (qi & 1) ? cs : sn;
  double cc = (qi & 1) ? sn : cs;
  if (qi & 2) ss = -ss;
  if ((qi + 1) & 2) cc = -cc;
  *s = ss; *c = cc;
}

float orc_tanhf(float x) {
  float a = fabsf(x);
  if (a >= 0.625f) {
    float r;
    if (a > 9.0f) r = 1.0f;
    else { float e = orc_expf(a + a); r = 1.0f - 2.0f / (e + 1.0f); }
    return x < 0.0f ? -r : r;
  }
  float z = x * x;
  float p = -5.70498872745e-3f;
  p = fmaf(p, z, 2.06390887954e-2f);
  p = fmaf(p, z, -5.37397155531e-2f);
  p = fmaf(p, z, 1.33314422036e-1f);
  p = fmaf(p, z, -3.33332819422e-1f);
  return fmaf(p * z, x, x);
}

/* ------------------------------------------------------------------ MLP ---
 * One stage of gymrl_mlp_forward (include/gymrl.h; the inference forward of
 * ActorCritic ppo_lunarlander.py:86-90, QNetwork dqn_cartpole.py:62-65, Actor
 * sac_pendulum.py:66-74): y = act(x W^T + b), W [out][in] row-major, in the kernel's
 * documented summation order — K padded to a multiple of 64, for each 16-block the
 * four k = kb+4q+j (q = 0..3) of MFMA j are chained with fmaf, j = 0..3 in turn.
 * act: 0 none, 1 tanh (orc_tanhf), 2 relu. */
void orc_linear_act(const float* x, const float* W, const float* b, int n, int in_dim, int out_dim, int act,
                    float* y) {
  const int k16 = (in_dim + 63) & ~63;
  for (int r = 0; r < n; ++r)
    for (int c = 0; c < out_dim; ++c) {
      float acc = 0.0f;
      for (int kb = 0; kb < k16; kb += 16)
        for (int j = 0; j < 4; ++j)
          for (int q = 0; q < 4; ++q) {
            const int k = kb + 4 * q + j;
            const float a = k < in_dim ? x[(size_t)r * in_dim + k] : 0.0f;
            const float w = k < in_dim ? W[(size_t)c * in_dim + k] : 0.0f;
            acc = fmaf(a, w, acc);
          }
      float v = acc + (b ? b[c] : 0.0f);
      if (act == 1) v = orc_tanhf(v);
      else if (act == 2) v = fmaxf(v, 0.0f);
      y[(size_t)r * out_dim + c] = v;
    }
}

/* ---------------------------------------------------- update-path GEMMs ---
 * The 256-wide layers of ActorCritic's training forward / backward (ppo_lunarlander.py:67-84,
 * :110-117, :303) in the accumulation order include/gymrl.h documents for gymrl_linear_fwd /
 * _bwd_input / _bwd_weight: one fmaf chain per output from +0; reduction index in chunks of 8
 * ascending, inside a chunk 0, 4, 1, 5, 2, 6, 3, 7. */
static const int kChunkOrder[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* Y [B, N] = X [B, K] W[N, K]^T + b (no activation) */
void orc_linear_fwd(const float* X, const float* W, const float* b, int B, int K, int N, float* Y) {
  for (int r = 0; r < B; ++r)
    for (int n = 0; n < N; ++n) {
      float acc = 0.0f;
      for (int c = 0; c < K; c += 8)
        for (int j = 0; j < 8; ++j) {
          const int k = c + kChunkOrder[j];
          acc = fmaf(X[(size_t)r * K + k], W[(size_t)n * K + k], acc);
        }
      Y[(size_t)r * N + n] = acc + (b ? b[n] : 0.0f);
    }
}

/* dX [B, K] = (dY [B, N] W[N, K]) * (1 - H^2)   (H NULL: no factor) */
void orc_linear_bwd_input(const float* dY, const float* W, const float* H, int B, int N, int K, float* dX) {
  for (int r = 0; r < B; ++r)
    for (int k = 0; k < K; ++k) {
      float acc = 0.0f;
      for (int c = 0; c < N; c += 8)
        for (int j = 0; j < 8; ++j) {
          const int n = c + kChunkOrder[j];
          acc = fmaf(dY[(size_t)r * N + n], W[(size_t)n * K + k], acc);
        }
      if (H) {
        const float h = H[(size_t)r * K + k];
        acc = acc * (1.0f - h * h);
      }
      dX[(size_t)r * K + k] = acc;
    }
}

/* dW [N, K] = dY^T X: rows in `slices` slices of `rps` rows, fmaf chain over the rows of a slice from +0,
 * slices combined as ((g0 + g1) + g2) + g3 in double, g_j = sum of slices s = j (mod 4) ascending. */
void orc_linear_bwd_weight(const float* dY, const float* X, int64_t B, int N, int K, int slices, int64_t rps,
                           float* dW) {
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      double g[4] = {0.0, 0.0, 0.0, 0.0};
      for (int s = 0; s < slices; ++s) {
        const int64_t m0 = (int64_t)s * rps, m1 = m0 + rps < B ? m0 + rps : B;
        float acc = 0.0f;
        for (int64_t m = m0; m < m1; ++m) acc = fmaf(dY[(size_t)m * N + n], X[(size_t)m * K + k], acc);
        g[s & 3] += (double)acc;
      }
      dW[(size_t)n * K + k] = (float)(((g[0] + g[1]) + g[2]) + g[3]);
    }
}

/* db [N] of gymrl_linear_bwd_weight: per slice the even-offset rows and the odd-offset rows are summed
 * sequentially in f32 and added (f32); slices combine like the weight tiles. */
void orc_linear_bwd_bias(const float* dY, int64_t B, int N, int slices, int64_t rps, float* db) {
  for (int n = 0; n < N; ++n) {
    double g[4] = {0.0, 0.0, 0.0, 0.0};
    for (int s = 0; s < slices; ++s) {
      const int64_t m0 = (int64_t)s * rps, m1 = m0 + rps < B ? m0 + rps : B;
      float ev = 0.0f, od = 0.0f;
      for (int64_t m = m0; m < m1; m += 2) ev += dY[(size_t)m * N + n];
      for (int64_t m = m0 + 1; m < m1; m += 2) od += dY[(size_t)m * N + n];
      g[s & 3] += (double)(ev + od);
    }
    db[n] = (float)(((g[0] + g[1]) + g[2]) + g[3]);
  }
}

/* ============================================================== Philox ==== */
/* Philox4x32-10 (Salmon et al. 2011), the build's own counter-based env/policy
 * stream; integer only. */
void orc_philox(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static float u01f_open0(uint32_t x) { return (float)((x >> 8) + 1u) * 0x1p-24f; }
static double u01d(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}
#define RNG_ENV_RESET 0x10000000u
#define RNG_ENV_STEP  0x20000000u
#define RNG_POLICY    0x30000000u
#define RNG_REPLAY    0x40000000u
#define RNG_NOISE     0x50000000u

/* ================================================================= GAE ==== */
/* G1 — PPOTrainer.compute_gae, ppo_lunarlander.py:179-196.  float64, env by env. */
void orc_gae(const float* rew, const float* val, const uint8_t* done, const float* next_val,
             int T, int N, double gamma, double lam, float* adv_out, float* ret_out,
             double* moments_out) {
  double s1 = 0.0, s2 = 0.0;
  /* :181 dones is a float32 array, so in :192 `gamma * gae_lambda * (1 - dones[t])` is
   * python-float * np.float32 = a FLOAT32 product under NumPy >= 2 (NEP 50 weak scalars);
   * in :189 gamma first meets a float64 (values[t+1]) so that product stays float64. */
  const float gl32 = (float)(gamma * lam);
  for (int n = 0; n < N; ++n) {
    double last = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      double vnext = (t == T - 1) ? (double)next_val[n] : (double)val[(size_t)(t + 1) * N + n];
      float nd32 = 1.0f - (done[(size_t)t * N + n] ? 1.0f : 0.0f);
      double nd = (double)nd32;
      double v = (double)val[(size_t)t * N + n];
      double delta = (double)rew[(size_t)t * N + n] + gamma * vnext * nd - v;   /* :188-190 */
      last = delta + (double)(gl32 * nd32) * last;                              /* :191-193 */
      adv_out[(size_t)t * N + n] = (float)last;
      ret_out[(size_t)t * N + n] = (float)(last + v);                            /* :195 */
      s1 += last; s2 += last * last;
    }
  }
  if (moments_out) { moments_out[0] = (double)T * (double)N; moments_out[1] = s1; moments_out[2] = s2; }
}

/* G2 — ReplayBuffer_on_policy.compute_advantage, utils/buffer.py:21-35.  float32. */
void orc_gae_dw(const float* rew, const float* val, const float* next_val, const uint8_t* done,
                const uint8_t* dw, int T, int N, double gamma, double lam, float* adv_out,
                float* vt_out, double* moments_out) {
  const float g32 = (float)gamma;           /* python scalar * f32 tensor -> f32 */
  const float gl32 = (float)(gamma * lam);  /* (gamma*lamda) is a python double, then weak-cast */
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < N; ++n) {
    float gae = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
      size_t o = (size_t)t * N + n;
      float delta = rew[o] + g32 * next_val[o] * (1.0f - (dw[o] ? 1.0f : 0.0f)) - val[o];  /* :23 */
      gae = gl32 * gae * (1.0f - (done[o] ? 1.0f : 0.0f)) + delta;                        /* :28 */
      adv_out[o] = gae;
      vt_out[o] = gae + val[o];                                                            /* :32 */
      s1 += (double)gae; s2 += (double)gae * (double)gae;
    }
  }
  if (moments_out) { moments_out[0] = (double)T * (double)N; moments_out[1] = s1; moments_out[2] = s2; }
}

/* G3 — compute_advantages, ppo_full_lunarlander.py:507-535. */
void orc_gae_decoupled(const float* rew, const float* val, const uint8_t* done,
                       const float* next_val, int T, int N, double gamma, double lam_actor,
                       double lam_critic, float* adv_actor_out, float* ret_out) {
  for (int n = 0; n < N; ++n) {
    double la = 0.0, lc = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      size_t o = (size_t)t * N + n;
      double vnext = (t == T - 1) ? (double)next_val[n] : (double)val[o + N];
      double nd = 1.0 - (done[o] ? 1.0 : 0.0);
      double v = (double)val[o];
      double delta = (double)rew[o] + gamma * vnext * nd - v;
      la = delta + gamma * lam_actor * nd * la;
      lc = delta + gamma * lam_critic * nd * lc;
      adv_actor_out[o] = (float)la;
      ret_out[o] = (float)(lc + v);
    }
  }
}

/* P5 — (A - mean) / (std + eps): ppo_lunarlander.py:236 (ddof 0), utils/buffer.py:33 (ddof 1). */
void orc_moments(const float* x, int64_t n, double* moments_out) {
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = 0; i < n; ++i) { s1 += (double)x[i]; s2 += (double)x[i] * (double)x[i]; }
  moments_out[0] = (double)n; moments_out[1] = s1; moments_out[2] = s2;
}
void orc_normalize(float* x, int64_t n, const double* moments, int ddof, double eps) {
  double cnt = moments[0], mean = moments[1] / cnt;
  double var = (moments[2] - cnt * mean * mean) / (cnt - (double)ddof);
  if (var < 0.0) var = 0.0;
  double denom = sqrt(var) + eps;
  for (int64_t i = 0; i < n; ++i) x[i] = (float)(((double)x[i] - mean) / denom);
}

/* ======================================================= categorical ====== */
static void log_softmax(const float* z, int A, float* ln, float* p, float* H) {
  float m = z[0];
  for (int k = 1; k < A; ++k) m = fmaxf(m, z[k]);
  float e[8], s = 0.0f;
  for (int k = 0; k < A; ++k) { e[k] = orc_expf(z[k] - m); s += e[k]; }
  float lse = m + orc_logf(s);
  float h = 0.0f;
  for (int k = 0; k < A; ++k) { ln[k] = z[k] - lse; p[k] = e[k] / s; h -= p[k] * ln[k]; }
  *H = h;
}

/* P2 — ActorCritic.get_action, ppo_lunarlander.py:92-104: Categorical(logits);
 * sample == argmax(p / q), q ~ Exp(1) (torch.multinomial CPU path). */
void orc_categorical_sample(const float* logits, const float* value_in, const float* noise_exp,
                            uint64_t seed, uint64_t counter, int64_t env_id0, int n, int A,
                            int deterministic, int32_t* act_out, float* logp_out, float* ent_out,
                            float* value_out) {
  for (int i = 0; i < n; ++i) {
    const float* z = logits + (size_t)i * A;
    float ln[8], p[8], H, q[8];
    log_softmax(z, A, ln, p, &H);
    int a = 0;
    if (deterministic) {
      for (int k = 1; k < A; ++k) if (z[k] > z[a]) a = k;            /* :98-99 argmax */
    } else {
      if (noise_exp) {
        for (int k = 0; k < A; ++k) q[k] = noise_exp[(size_t)i * A + k];
      } else {
        uint64_t env = (uint64_t)(env_id0 + i);
        for (int blk = 0; blk < (A + 3) / 4; ++blk) {
          uint32_t w[4];
          orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), (uint32_t)counter,
                     RNG_POLICY | ((uint32_t)((counter >> 32) & 0x3FFFFFu) << 2) | (uint32_t)blk, w);
          for (int k = 0; k < 4 && blk * 4 + k < A; ++k) q[blk * 4 + k] = -orc_logf(u01f_open0(w[k]));
        }
      }
      float best = p[0] / q[0];
      for (int k = 1; k < A; ++k) { float c = p[k] / q[k]; if (c > best) { best = c; a = k; } }
    }
    act_out[i] = a;
    logp_out[i] = ln[a];                                              /* :103 log_prob */
    if (ent_out) ent_out[i] = H;
    if (value_out && value_in) value_out[i] = value_in[i];
  }
}

/* ============================================================ PPO loss ==== */
typedef struct { float clip_eps, dual_clip, value_coef, entropy_coef; } orc_ppo_cfg;

/* L1+L2 — evaluate_actions + loss + metrics, ppo_lunarlander.py:110-117, 278-322.
 * Gradients restate torch autograd (min/max ties split 1/2, clamp passes grad on
 * the closed interval). */
void orc_ppo_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                          const int32_t* act, const float* logp_old, const float* adv,
                          const float* ret, const double* adv_moments, int B, int A,
                          const orc_ppo_cfg* cfg, float* dlogits_out, float* dvalue_out,
                          double* metrics_sum) {
  double met[5] = {0, 0, 0, 0, 0};
  const float invB = 1.0f / (float)B;
  const float lo = 1.0f - cfg->clip_eps, hi = 1.0f + cfg->clip_eps;
  for (int b = 0; b < B; ++b) {
    const float* z = logits + (size_t)b * A;
    int i = idx ? idx[b] : b;
    int a = act[i];
    float ad = adv[i];
    if (adv_moments) {                                                /* :236 */
      double cnt = adv_moments[0], mean = adv_moments[1] / cnt;
      double var = adv_moments[2] / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      ad = (float)(((double)ad - mean) / (sqrt(var) + 1e-8));
    }
    float ln[8], p[8], H;
    log_softmax(z, A, ln, p, &H);
    float lp = ln[a];
    float ratio = orc_expf(lp - logp_old[i]);                         /* :278 */
    float s1 = ratio * ad;                                            /* :279 */
    float rc = fminf(fmaxf(ratio, lo), hi);
    float s2 = rc * ad;                                               /* :280-283 */
    float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
    float ms = fminf(s1, s2);                                         /* :285 */
    float dms_dr = w1 * ad + (1.0f - w1) * ad * inr;
    float obj = ms;
    if (ad < 0.0f) {                                                  /* :287-291 dual clip */
      float dc = cfg->dual_clip * ad;
      obj = fmaxf(ms, dc);
      float wm = ms > dc ? 1.0f : (ms == dc ? 0.5f : 0.0f);
      dms_dr *= wm;
    }
    float g_lp = -invB * dms_dr * ratio;
    float g_H = -cfg->entropy_coef * invB;                            /* :298 */
    for (int k = 0; k < A; ++k) {
      float onehot = (a == k) ? 1.0f : 0.0f;
      dlogits_out[(size_t)b * A + k] = g_lp * (onehot - p[k]) + g_H * (-p[k] * (ln[k] + H));
    }
    float dvr = value[b] - ret[i];
    dvalue_out[b] = cfg->value_coef * 2.0f * dvr * invB;              /* :294-296 */
    met[0] += -(double)obj;
    met[1] += (double)(cfg->value_coef * (dvr * dvr));
    met[2] += (double)H;
    met[3] += (ratio < lo || ratio > hi) ? 1.0 : 0.0;                 /* :313-318 */
    met[4] += (double)(logp_old[i] - lp);                             /* :320 */
  }
  if (metrics_sum) for (int k = 0; k < 5; ++k) metrics_sum[k] += met[k];
}

typedef struct {
  float clip_eps_min, clip_eps_max, dual_clip, erc_beta_low, erc_beta_high, entropy_coef;
} orc_ppo_full_cfg;

/* L3 — update_model minibatch, ppo_full_lunarlander.py:575-652 (clip_cov_ratio = 0). */
void orc_ppo_full_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                               const int32_t* act, const float* logp_old, const float* ent_old,
                               const float* adv, const float* ret, int B, int A,
                               const orc_ppo_full_cfg* cfg, const float* corr_mul, float* dlogits_out,
                               float* dvalue_out, double* metrics_sum) {
  double met[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float invB = 1.0f / (float)B;
  const float lo = 1.0f - cfg->clip_eps_min, hi = 1.0f + cfg->clip_eps_max;
  for (int b = 0; b < B; ++b) {
    const float* z = logits + (size_t)b * A;
    int i = idx ? idx[b] : b;
    int a = act[i];
    float ad = adv[i];
    float ln[8], p[8], H;
    log_softmax(z, A, ln, p, &H);
    float lp = ln[a];
    float er = H / (ent_old[i] + 1e-8f);                              /* :586 */
    float erc = (er > (1.0f - cfg->erc_beta_low) && er < (1.0f + cfg->erc_beta_high)) ? 1.0f : 0.0f;
    float corr = erc;
    if (corr_mul) corr *= corr_mul[b];                                /* covariance clip :611-616 / :747-753 */
    float ratio = orc_expf(lp - logp_old[i]);                         /* :593 */
    float r1 = fminf(fmaxf(ratio, 0.0f), cfg->dual_clip);             /* :600 */
    float r2 = fminf(fmaxf(ratio, lo), hi);                           /* :603-607 */
    float s1 = r1 * ad, s2 = r2 * ad;
    float in1 = (ratio >= 0.0f && ratio <= cfg->dual_clip) ? 1.0f : 0.0f;
    float in2 = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
    float ms = fminf(s1, s2);
    float dms_dr = w1 * ad * in1 + (1.0f - w1) * ad * in2;
    float g_lp = -invB * corr * dms_dr * ratio;                       /* :624 */
    float g_H = -cfg->entropy_coef * invB * corr;                     /* :632-633 */
    for (int k = 0; k < A; ++k) {
      float onehot = (a == k) ? 1.0f : 0.0f;
      dlogits_out[(size_t)b * A + k] = g_lp * (onehot - p[k]) + g_H * (-p[k] * (ln[k] + H));
    }
    float dvr = value[b] - ret[i];
    dvalue_out[b] = corr * dvr * invB;                                /* :627-629 */
    met[0] += (double)(-ms * corr);
    met[1] += (double)(0.5f * corr * (dvr * dvr));
    met[2] += (double)(H * corr);
    met[3] += (ratio < lo || ratio > hi) ? (double)corr : 0.0;        /* :617-623 */
    met[4] += (double)(logp_old[i] - lp);
    met[5] += 1.0 - (double)erc;                                     /* :652 */
    met[6] += (double)lp; met[7] += (double)ad; met[8] += (double)lp * (double)ad;
  }
  if (metrics_sum) for (int k = 0; k < 9; ++k) metrics_sum[k] += met[k];
}

/* L4 — recurrent PPO minibatch, ppo_lstm_lunarlander.py:716-776: L3 with masked means (:646-655)
 * and the clipped value loss (:763-770).  metrics_sum f64[10] (last = sum corr). */
void orc_ppo_rnn_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                              const int32_t* act, const float* logp_old, const float* ent_old,
                              const float* val_old, const float* adv, const float* ret, int B, int A,
                              const orc_ppo_full_cfg* cfg, const float* corr_mul, float* dlogits_out,
                              float* dvalue_out, double* metrics_sum) {
  double met[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float lo = 1.0f - cfg->clip_eps_min, hi = 1.0f + cfg->clip_eps_max;
  unsigned cnt = 0;
  for (int b = 0; b < B; ++b) {                                       /* erc_mask :722-726 */
    float ln[8], p[8], H;
    log_softmax(logits + (size_t)b * A, A, ln, p, &H);
    float er = H / (ent_old[idx ? idx[b] : b] + 1e-8f);
    cnt += (er > (1.0f - cfg->erc_beta_low) && er < (1.0f + cfg->erc_beta_high) && (!corr_mul || corr_mul[b] != 0.0f)) ? 1u : 0u;
  }
  const float inv = cnt ? 1.0f / (float)cnt : 0.0f;                   /* masked_mean: sum / count, 0 if empty */
  for (int b = 0; b < B; ++b) {
    const float* z = logits + (size_t)b * A;
    int i = idx ? idx[b] : b;
    int a = act[i];
    float ad = adv[i], v = value[b], vo = val_old[i], rt = ret[i];
    float ln[8], p[8], H;
    log_softmax(z, A, ln, p, &H);
    float lp = ln[a];
    float er = H / (ent_old[i] + 1e-8f);
    float erc = (er > (1.0f - cfg->erc_beta_low) && er < (1.0f + cfg->erc_beta_high)) ? 1.0f : 0.0f;
    float corr = erc;
    if (corr_mul) corr *= corr_mul[b];                                /* covariance clip :611-616 / :747-753 */
    float ratio = orc_expf(lp - logp_old[i]);                         /* :728 */
    float r1 = fminf(fmaxf(ratio, 0.0f), cfg->dual_clip);             /* :734 */
    float r2 = fminf(fmaxf(ratio, lo), hi);                           /* :736-741 */
    float s1 = r1 * ad, s2 = r2 * ad;
    float in1 = (ratio >= 0.0f && ratio <= cfg->dual_clip) ? 1.0f : 0.0f;
    float in2 = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);
    float ms = fminf(s1, s2);
    float dms_dr = w1 * ad * in1 + (1.0f - w1) * ad * in2;
    float scale = corr * inv;
    float g_lp = -scale * dms_dr * ratio;                             /* :761 */
    float g_H = -cfg->entropy_coef * scale;                           /* :772-773 */
    for (int k = 0; k < A; ++k) {
      float onehot = (a == k) ? 1.0f : 0.0f;
      dlogits_out[(size_t)b * A + k] = g_lp * (onehot - p[k]) + g_H * (-p[k] * (ln[k] + H));
    }
    float dv = v - vo;                                                /* :763-770 */
    float vc = vo + fminf(fmaxf(dv, -cfg->clip_eps_min), cfg->clip_eps_max);
    float inr = (dv >= -cfg->clip_eps_min && dv <= cfg->clip_eps_max) ? 1.0f : 0.0f;
    float e1 = v - rt, e2 = vc - rt;
    float l1 = e1 * e1, l2 = e2 * e2;
    float wv = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
    dvalue_out[b] = 0.5f * scale * (wv * 2.0f * e1 + (1.0f - wv) * 2.0f * e2 * inr);
    met[0] += (double)(-ms * corr);
    met[1] += (double)(0.5f * corr * fmaxf(l1, l2));
    met[2] += (double)(H * corr);
    met[3] += (ratio < lo || ratio > hi) ? (double)corr : 0.0;
    met[4] += (double)(logp_old[i] - lp);
    met[5] += 1.0 - (double)erc;
    met[6] += (double)lp; met[7] += (double)ad; met[8] += (double)lp * (double)ad;
    met[9] += (double)corr;
  }
  if (metrics_sum) for (int k = 0; k < 10; ++k) metrics_sum[k] += met[k];
}

/* URNN's GRU cell, pointwise half — ppo_lstm_lunarlander.py:449-491 (torch.nn.GRU gate order r, z, n). */
static float orc_sigmoidf(float x) { return 1.0f / (1.0f + orc_expf(-x)); }
void orc_gru_cell_fwd(const float* gi, const float* gh, const float* h, int B, int H, float* h_out) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < H; ++c) {
      const float* i3 = gi + (size_t)b * 3 * H; const float* h3 = gh + (size_t)b * 3 * H;
      float r = orc_sigmoidf(i3[c] + h3[c]);
      float z = orc_sigmoidf(i3[H + c] + h3[H + c]);
      float n = orc_tanhf(i3[2 * H + c] + r * h3[2 * H + c]);
      h_out[(size_t)b * H + c] = (1.0f - z) * n + z * h[(size_t)b * H + c];
    }
}
void orc_gru_cell_bwd(const float* gi, const float* gh, const float* h, const float* dh_out, int B, int H,
                      float* dgi, float* dgh, float* dh) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < H; ++c) {
      size_t g0 = (size_t)b * 3 * H + c, h0 = (size_t)b * H + c;
      float r = orc_sigmoidf(gi[g0] + gh[g0]);
      float z = orc_sigmoidf(gi[g0 + H] + gh[g0 + H]);
      float hn = gh[g0 + 2 * H];
      float n = orc_tanhf(gi[g0 + 2 * H] + r * hn);
      float go = dh_out[h0];
      float dn = go * (1.0f - z), dz = go * (h[h0] - n);
      float dnp = dn * (1.0f - n * n);
      float dr = (dnp * hn) * (r * (1.0f - r)), dzp = dz * (z * (1.0f - z));
      dgi[g0] = dr; dgi[g0 + H] = dzp; dgi[g0 + 2 * H] = dnp;
      dgh[g0] = dr; dgh[g0 + H] = dzp; dgh[g0 + 2 * H] = dnp * r;
      dh[h0] = go * z;
    }
}

/* RND reward — :588-590: float32 mean of squares; summation order of the device kernel (64 strided
 * partial sums, then a halving tree). */
void orc_rnd_reward(const float* predict, const float* target, int B, int E, float* rew_inout, float* rnd_out) {
  for (int b = 0; b < B; ++b) {
    float part[64];
    for (int l = 0; l < 64; ++l) {
      float s = 0.0f;
      for (int c = l; c < E; c += 64) { float d = predict[(size_t)b * E + c] - target[(size_t)b * E + c]; s += d * d; }
      part[l] = s;
    }
    for (int off = 32; off > 0; off >>= 1) for (int l = 0; l < off; ++l) part[l] += part[l + off];
    float m = part[0] / (float)E;
    if (rnd_out) rnd_out[b] = m;
    if (rew_inout) rew_inout[b] = rew_inout[b] + m;
  }
}

/* P6 epoch shuffle — ppo_lunarlander.py:262: the keyed bijection gymrl_permutation evaluates (6 alternating
 * Feistel rounds with a Philox round function, cycle-walked into [0, M)). */
static uint32_t feistel_once(uint32_t x, int a, int b, uint64_t seed, uint64_t counter) {
  const uint32_t mask_lo = (1u << a) - 1u, mask_hi = (1u << b) - 1u;
  uint32_t lo = x & mask_lo, hi = x >> a, o[4];
  const uint32_t c2 = (uint32_t)counter, c3 = 0x60000000u | ((uint32_t)(counter >> 32) & 0x0FFFFFFFu);
  for (uint32_t r = 0; r < 6; ++r) {
    if ((r & 1u) == 0u) { orc_philox(seed, hi, r, c2, c3, o); lo ^= o[0] & mask_lo; }
    else                { orc_philox(seed, lo, r, c2, c3, o); hi ^= o[0] & mask_hi; }
  }
  return (hi << a) | lo;
}
void orc_permutation(uint64_t seed, uint64_t counter, int64_t M, int32_t* out) {
  int bits = 2;
  while (((int64_t)1 << bits) < M) ++bits;
  const int a = bits / 2, b = bits - a;
  for (int64_t i = 0; i < M; ++i) {
    uint32_t x = feistel_once((uint32_t)i, a, b, seed, counter);
    while (x >= (uint32_t)M) x = feistel_once(x, a, b, seed, counter);
    out[i] = (int32_t)x;
  }
}

/* =========================================================== optimiser ==== */
/* O1 — clip_grad_norm_ + torch.optim.Adam step, ppo_lunarlander.py:169,302-307. */
/* The squared gradient norm in gymrl_sqnorm's documented order (include/gymrl.h; optim.hip sqnorm_partial_kernel /
 * sqnorm_final_kernel), so that the clip coefficient — and with it every parameter after the step — is pinned bit for
 * bit, not to 1e-6: nb = clamp(ceil(n / 4096), 1, 1024) workgroups of 256 threads; thread t of workgroup b adds the
 * float4 groups b*256 + t, + nb*256, ... as ((a^2 + b^2) + c^2) + d^2 in float64 (the products of two float32 are
 * exact); the n % 4 tail elements go to threads 0..2 of workgroup 0; each 64-lane wave folds by halves
 * (v[l] += v[l + off], off = 32, 16, .. 1), the four wave sums are added in order; the final pass gives thread t the
 * partials t, t + 256, ... in order and folds the 256 sums by halves. */
void orc_sqnorm(const float* g, int64_t n, float grad_scale, double* out) {
  int64_t nb = (n + 4095) / 4096;
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  const int64_t n4 = n >> 2, stride = nb * 256;
  double fin[256];
  for (int t = 0; t < 256; ++t) fin[t] = 0.0;
  for (int64_t b = 0; b < nb; ++b) {
    double lane[256];
    for (int t = 0; t < 256; ++t) {
      double s = 0.0;
      for (int64_t i = b * 256 + t; i < n4; i += stride) {
        float a = g[4 * i] * grad_scale, bb = g[4 * i + 1] * grad_scale, c = g[4 * i + 2] * grad_scale, d = g[4 * i + 3] * grad_scale;
        s += (double)a * (double)a + (double)bb * (double)bb + (double)c * (double)c + (double)d * (double)d;
      }
      if (b == 0 && t < (int)(n & 3)) { float a = g[(n4 << 2) + t] * grad_scale; s += (double)a * (double)a; }
      lane[t] = s;
    }
    double part = 0.0;
    for (int w = 0; w < 4; ++w) {
      double* v = lane + 64 * w;
      for (int off = 32; off > 0; off >>= 1) for (int l = 0; l < off; ++l) v[l] += v[l + off];
      part += v[0];
    }
    fin[b & 255] += part;                        /* thread (b % 256) of the final pass meets its partials in order of b */
  }
  for (int s2 = 128; s2 > 0; s2 >>= 1) for (int t = 0; t < s2; ++t) fin[t] += fin[t + s2];
  out[0] = fin[0];
}
void orc_adam_step(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1,
                   double beta2, double eps, int64_t step, float grad_scale, float max_grad_norm,
                   const double* sqnorm, float clamp_abs, int zero_grad) {
  float scale = 1.0f;
  if (max_grad_norm > 0.0f && sqnorm) {
    float total = (float)sqrt(sqnorm[0]);
    float coef = max_grad_norm / (total + 1e-6f);
    scale = coef < 1.0f ? coef : 1.0f;
  }
  double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  float omb1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2), e = (float)eps;
  for (int64_t i = 0; i < n; ++i) {
    float gg = g[i] * grad_scale;
    gg = gg * scale;
    if (clamp_abs > 0.0f) gg = fminf(fmaxf(gg, -clamp_abs), clamp_abs);   /* dqn_cartpole.py:163-165 */
    m[i] = m[i] + (gg - m[i]) * omb1;
    v[i] = v[i] * b2 + omb2 * gg * gg;
    float denom = sqrtf(v[i]) / bc2_sqrt + e;
    p[i] = p[i] - step_size * (m[i] / denom);
    if (zero_grad) g[i] = 0.0f;
  }
}
/* R4/A4 — rainbow_dqn_cartpole.py:347-352, sac_pendulum.py:194-199. */
void orc_soft_update(float* target, const float* source, int64_t n, double tau) {
  float t = (float)tau, omt = (float)(1.0 - tau);
  for (int64_t i = 0; i < n; ++i) target[i] = t * source[i] + omt * target[i];
}

/* ================================================================ envs ==== */
/* gymnasium classic_control restated (third-party; unpinned — see header). */
enum { ORC_CARTPOLE = 0, ORC_PENDULUM = 1, ORC_LUNARLANDER = 2 };

typedef struct {
  double s[4];        /* cartpole: x, xdot, th, thdot ; pendulum: th, thdot */
  double ep_ret;
  int32_t ep_len;
  uint32_t episode;
} orc_classic_env;

typedef struct {
  int kind, n;
  uint64_t seed;
  int64_t env_id0;
  orc_classic_env* classic;
  void* lunar;
} orc_env;

void* orc_lunar_alloc(int n);
void orc_lunar_reset_one(void* st, int i, uint64_t seed, uint64_t env, uint32_t episode, float* obs);
void orc_lunar_step_one(void* st, int i, uint64_t seed, uint64_t env, int action, float* obs_next,
                        float* obs_term, float* rew, uint8_t* terminated, uint8_t* truncated,
                        int* done, double* ep_ret, int* ep_len);

orc_env* orc_env_create(int kind, int n, uint64_t seed, int64_t env_id0) {
  orc_env* e = (orc_env*)calloc(1, sizeof(orc_env));
  e->kind = kind; e->n = n; e->seed = seed; e->env_id0 = env_id0;
  if (kind == ORC_LUNARLANDER) e->lunar = orc_lunar_alloc(n);
  else e->classic = (orc_classic_env*)calloc((size_t)n, sizeof(orc_classic_env));
  return e;
}
void orc_env_destroy(orc_env* e) { if (e) { free(e->classic); free(e->lunar); free(e); } }
void orc_lunar_get_words(void* st, int i, uint32_t* out144);
void orc_env_lunar_words(orc_env* e, int i, uint32_t* out144) { orc_lunar_get_words(e->lunar, i, out144); }

/* Test hook (tests/classic_micro.py): put env i into a given float64 state (CartPole: x, xdot, th, thdot; Pendulum: th,
 * thdot) with `ep_len` steps of its episode already taken — the closed-form scenarios start from hand-picked states. */
void orc_env_set_classic(orc_env* e, int i, const double* s, int ep_len) {
  if (e->kind == ORC_LUNARLANDER || i < 0 || i >= e->n) return;
  const int k = e->kind == ORC_CARTPOLE ? 4 : 2;
  for (int j = 0; j < k; ++j) e->classic[i].s[j] = s[j];
  e->classic[i].ep_len = ep_len;
  e->classic[i].ep_ret = 0.0;
}

static void cartpole_draw(uint64_t seed, uint64_t env, uint32_t episode, double* s) {
  uint32_t a[4], b[4];
  orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 0u, a);
  orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 1u, b);
  s[0] = -0.05 + 0.1 * u01d(a[0], a[1]);
  s[1] = -0.05 + 0.1 * u01d(a[2], a[3]);
  s[2] = -0.05 + 0.1 * u01d(b[0], b[1]);
  s[3] = -0.05 + 0.1 * u01d(b[2], b[3]);
}
static void pendulum_draw(uint64_t seed, uint64_t env, uint32_t episode, double* s) {
  uint32_t a[4];
  const double pi = 3.14159265358979323846;
  orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 0u, a);
  s[0] = -pi + (2.0 * pi) * u01d(a[0], a[1]);
  s[1] = -1.0 + 2.0 * u01d(a[2], a[3]);
}
static void classic_obs(int kind, const double* s, float* o) {
  if (kind == ORC_CARTPOLE) { for (int k = 0; k < 4; ++k) o[k] = (float)s[k]; }
  else { double sn, cs; orc_sincos(s[0], &sn, &cs); o[0] = (float)cs; o[1] = (float)sn; o[2] = (float)s[1]; }
}

void orc_env_reset(orc_env* e, float* obs_out) {
  int D = e->kind == ORC_CARTPOLE ? 4 : e->kind == ORC_PENDULUM ? 3 : 8;
  for (int i = 0; i < e->n; ++i) {
    uint64_t env = (uint64_t)(e->env_id0 + i);
    if (e->kind == ORC_LUNARLANDER) { orc_lunar_reset_one(e->lunar, i, e->seed, env, 0u, obs_out + (size_t)i * D); continue; }
    orc_classic_env* c = &e->classic[i];
    memset(c, 0, sizeof(*c));
    if (e->kind == ORC_CARTPOLE) cartpole_draw(e->seed, env, 0u, c->s); else pendulum_draw(e->seed, env, 0u, c->s);
    classic_obs(e->kind, c->s, obs_out + (size_t)i * D);
  }
}

/* One vector step with auto-reset (semantics of gymrl_env_step in include/gymrl.h). */
/* gymrl_env_abandon: classic envs whose running episode reached `cap` steps start the next one. */
void orc_env_abandon(orc_env* e, int cap, float* obs_inout, uint8_t* flag_inout, float* ep_ret_out, int32_t* ep_len_out) {
  const int D = e->kind == ORC_CARTPOLE ? 4 : 3;
  if (e->kind == ORC_LUNARLANDER) return;
  for (int i = 0; i < e->n; ++i) {
    orc_classic_env* c = &e->classic[i];
    if (c->ep_len < cap) continue;
    if (ep_ret_out) ep_ret_out[i] = (float)c->ep_ret;
    if (ep_len_out) ep_len_out[i] = c->ep_len;
    if (flag_inout) flag_inout[i] |= 1;
    c->episode += 1u; c->ep_ret = 0.0; c->ep_len = 0;
    if (e->kind == ORC_CARTPOLE) cartpole_draw(e->seed, (uint64_t)(e->env_id0 + i), c->episode, c->s);
    else pendulum_draw(e->seed, (uint64_t)(e->env_id0 + i), c->episode, c->s);
    classic_obs(e->kind, c->s, obs_inout + (size_t)i * D);
  }
}

void orc_env_step(orc_env* e, const void* action, float* obs_out, float* term_obs_out,
                  float* rew_out, uint8_t* terminated_out, uint8_t* truncated_out,
                  uint8_t* done_out, float* ep_ret_out, int32_t* ep_len_out, double* ep_stats) {
  const int D = e->kind == ORC_CARTPOLE ? 4 : e->kind == ORC_PENDULUM ? 3 : 8;
  for (int i = 0; i < e->n; ++i) {
    uint64_t env = (uint64_t)(e->env_id0 + i);
    float o_term[8], o_next[8];
    int done = 0, len = 0; double ret = 0.0;
    if (e->kind == ORC_LUNARLANDER) {
      orc_lunar_step_one(e->lunar, i, e->seed, env, ((const int32_t*)action)[i], o_next, o_term,
                         &rew_out[i], &terminated_out[i], &truncated_out[i], &done, &ret, &len);
    } else {
      orc_classic_env* c = &e->classic[i];
      int terminated = 0, truncated = 0; double reward;
      if (e->kind == ORC_CARTPOLE) {
        double x = c->s[0], xd = c->s[1], th = c->s[2], thd = c->s[3];
        double force = ((const int32_t*)action)[i] == 1 ? 10.0 : -10.0;
        double co, si;
        orc_sincos(th, &si, &co);
        double temp = (force + 0.05 * (thd * thd) * si) / 1.1;
        double thacc = (9.8 * si - co * temp) / (0.5 * (4.0 / 3.0 - 0.1 * (co * co) / 1.1));
        double xacc = temp - 0.05 * thacc * co / 1.1;
        x = x + 0.02 * xd; xd = xd + 0.02 * xacc; th = th + 0.02 * thd; thd = thd + 0.02 * thacc;
        c->s[0] = x; c->s[1] = xd; c->s[2] = th; c->s[3] = thd;
        double lim = 12.0 * 2.0 * 3.14159265358979323846 / 360.0;
        terminated = x < -2.4 || x > 2.4 || th < -lim || th > lim;
        reward = 1.0;
        len = c->ep_len + 1; truncated = len >= 500;
      } else {
        const double pi = 3.14159265358979323846;
        double th = c->s[0], thd = c->s[1];
        double u = (double)((const float*)action)[i];
        u = u < -2.0 ? -2.0 : (u > 2.0 ? 2.0 : u);
        double a = th + pi; a = a - floor(a / (2.0 * pi)) * (2.0 * pi);
        double an = a - pi;
        double cost = an * an + 0.1 * (thd * thd) + 0.001 * (u * u);
        double sin_th, cos_th;
        orc_sincos(th, &sin_th, &cos_th);
        (void)cos_th;
        double nthd = thd + (15.0 * sin_th + 3.0 * u) * 0.05;
        nthd = nthd < -8.0 ? -8.0 : (nthd > 8.0 ? 8.0 : nthd);
        c->s[0] = th + nthd * 0.05; c->s[1] = nthd;
        reward = -cost;
        len = c->ep_len + 1; truncated = len >= 200;
      }
      done = terminated || truncated;
      ret = c->ep_ret + reward;
      rew_out[i] = (float)reward;
      terminated_out[i] = (uint8_t)terminated; truncated_out[i] = (uint8_t)truncated;
      classic_obs(e->kind, c->s, o_term);
      if (done) {
        uint32_t ep = c->episode + 1u;
        if (e->kind == ORC_CARTPOLE) cartpole_draw(e->seed, env, ep, c->s); else pendulum_draw(e->seed, env, ep, c->s);
        c->ep_ret = 0.0; c->ep_len = 0; c->episode = ep;
        classic_obs(e->kind, c->s, o_next);
      } else {
        c->ep_ret = ret; c->ep_len = len;
        memcpy(o_next, o_term, sizeof(float) * D);
      }
    }
    if (done_out) done_out[i] = (uint8_t)done;
    memcpy(obs_out + (size_t)i * D, o_next, sizeof(float) * D);
    if (term_obs_out) memcpy(term_obs_out + (size_t)i * D, o_term, sizeof(float) * D);
    if (done) {
      if (ep_ret_out) ep_ret_out[i] = (float)ret;
      if (ep_len_out) ep_len_out[i] = len;
      if (ep_stats) { ep_stats[0] += 1.0; ep_stats[1] += ret; ep_stats[2] += (double)len; }
    }
  }
}
