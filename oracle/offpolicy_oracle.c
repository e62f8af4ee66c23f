/*
 * offpolicy_oracle.c — CPU restatement of the DQN / Rainbow / SAC learner-side pieces and
 * utils/normalization.py.  TEST INFRASTRUCTURE ONLY (see gymrl_oracle.c header).
 * Pinned by tests/test_oracle_golden_offpolicy.py against vectors captured from the
 * reference's own classes (SumTree, PrioritizedNStepBuffer, NoisyLinear, Actor.sample,
 * SACTrainer.update, DQN/Rainbow update, RunningMeanStd/Normalization/RewardScaling).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float orc_expf(float x);
float orc_logf(float x);
float orc_tanhf(float x);
void orc_sincosf(float x, float* s, float* c);
void orc_philox(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
#define RNG_POLICY 0x30000000u
#define RNG_REPLAY 0x40000000u
#define RNG_NOISE  0x50000000u
static float u01f(uint32_t x) { return (float)(x >> 8) * 0x1p-24f; }
static float u01f_open0(uint32_t x) { return (float)((x >> 8) + 1u) * 0x1p-24f; }
static double u01d(uint32_t a, uint32_t b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53; }

/* ============================================================ SumTree ===== */
/* S1 — rainbow_dqn_cartpole.py:116-152 (and ddqn_per_cartpole.py:67-104). */
void orc_tree_update(double* tree, int64_t cap, int64_t data_index, double priority) {   /* :122-128 */
  int64_t t = data_index + cap - 1;
  double change = priority - tree[t];
  tree[t] = priority;
  while (t != 0) { t = (t - 1) / 2; tree[t] += change; }
}
/* The N-ROW VECTOR STORE (gymrl_per_update with idx == NULL; include/gymrl.h "vector store").  The reference stores ONE row
 * per env step (:201-205), so the order in which the N rows of a vector step reach an ancestor is this build's own
 * definition, chosen so that no node needs an N-long dependent chain of float64 adds:
 *   1. rows b = 0..B-1 go to leaves (idx_start + b) % cap (B <= cap: all distinct); change_b = p_b - old leaf; leaf := p_b;
 *   2. a complete binary tree over the batch index: seg[P + b] = change_b (+0.0 for b >= B, P = the power of two >= B),
 *      seg[k] = seg[2k] + seg[2k + 1];
 *   3. for every ancestor node: its elements (the b whose leaf lies below it) in ascending b, grouped into maximal runs
 *      of CONSECUTIVE b; a run [a, e) is summed from the canonical blocks of the batch tree
 *      (l = a + P, r = e + P; while l < r: if l odd: sl += seg[l++]; if r odd: sr += seg[--r]; l, r >>= 1; run = sl + sr,
 *      sl = sr = +0.0 initially); S = +0.0, S += run for the runs in ascending order; node := node + S.
 *   4. B > 8192: consecutive sub-stores of 8192 rows, one after the other.
 * At B = 1 this is node := node + change — the reference's SumTree.update (:122-128) bit for bit, so every fixture taken
 * from the reference's one-row stores (sumtree.npz, per_nstep.npz, rainbow_trace.npz) pins it.  Written here with explicit
 * member lists (no tree-position arithmetic): the HIP kernel finds the same runs in closed form. */
static double seg_run(const double* seg, int64_t P, int64_t a, int64_t e) {
  double sl = 0.0, sr = 0.0;
  for (int64_t l = a + P, r = e + P; l < r; l >>= 1, r >>= 1) {
    if (l & 1) sl += seg[l++];
    if (r & 1) sr += seg[--r];
  }
  return sl + sr;
}
static void tree_store_chunk(double* tree, int64_t cap, int64_t start, const double* prio, double prio_scalar, int B) {
  int64_t P = 1;
  while (P < B) P <<= 1;
  double* seg = (double*)calloc((size_t)(2 * P), sizeof(double));
  for (int b = 0; b < B; ++b) {
    int64_t t = (start + b) % cap + cap - 1;
    double p = prio ? prio[b] : prio_scalar;
    seg[P + b] = p - tree[t];
    tree[t] = p;
  }
  for (int64_t k = P - 1; k >= 1; --k) seg[k] = seg[2 * k] + seg[2 * k + 1];
  /* member lists: every ancestor met by the walks gets its elements in ascending b */
  int depth = 0;
  for (int64_t t = 2 * cap - 2; t > 0; t = (t - 1) / 2) ++depth;
  size_t cap_pairs = (size_t)B * (size_t)(depth + 1) + 1, n_pairs = 0;
  int64_t* pn = (int64_t*)malloc(cap_pairs * sizeof(int64_t));          /* (node, b) pairs in walk order */
  int32_t* pb = (int32_t*)malloc(cap_pairs * sizeof(int32_t));
  for (int b = 0; b < B; ++b)
    for (int64_t t = (start + b) % cap + cap - 1; t != 0;) { t = (t - 1) / 2; pn[n_pairs] = t; pb[n_pairs++] = b; }
  /* stable counting sort by node over the touched nodes: head[node] -> first pair, linked in ascending b */
  int64_t tcap = 2 * cap - 1;
  int64_t* first = (int64_t*)malloc((size_t)tcap * sizeof(int64_t));
  int64_t* last = (int64_t*)malloc((size_t)tcap * sizeof(int64_t));
  int64_t* next = (int64_t*)malloc(cap_pairs * sizeof(int64_t));
  for (int64_t t = 0; t < tcap; ++t) first[t] = -1;
  for (size_t k = 0; k < n_pairs; ++k) {
    next[k] = -1;
    if (first[pn[k]] < 0) first[pn[k]] = (int64_t)k; else next[last[pn[k]]] = (int64_t)k;
    last[pn[k]] = (int64_t)k;
  }
  for (size_t k0 = 0; k0 < n_pairs; ++k0) {
    int64_t node = pn[k0];
    if (first[node] != (int64_t)k0) continue;                          /* each node once, at its first pair */
    double S = 0.0;
    int64_t k = (int64_t)k0;
    while (k >= 0) {
      int64_t a = pb[k], e = a + 1;
      k = next[k];
      while (k >= 0 && pb[k] == e) { ++e; k = next[k]; }               /* a maximal run of consecutive b */
      S += seg_run(seg, P, a, e);
    }
    tree[node] = tree[node] + S;
  }
  free(seg); free(pn); free(pb); free(first); free(last); free(next);
}
void orc_tree_update_many(double* tree, int64_t cap, const int32_t* idx, int64_t idx_start, int idx_is_tree,
                          const double* prio, double prio_scalar, int B) {               /* :258-261 */
  if (!idx) {                                   /* the vector store (see above); B <= cap */
    for (int o = 0; o < B; o += 8192)
      tree_store_chunk(tree, cap, idx_start + o, prio ? prio + o : 0, prio_scalar, B - o < 8192 ? B - o : 8192);
    return;
  }
  for (int i = 0; i < B; ++i) {                 /* update_priorities: the reference's sequential loop, duplicates and all */
    int64_t d = idx_is_tree ? (int64_t)idx[i] - (cap - 1) : (int64_t)idx[i];
    orc_tree_update(tree, cap, d, prio ? prio[i] : prio_scalar);
  }
}
int64_t orc_tree_get_index(const double* tree, int64_t cap, double v, double* prio_out) {  /* :130-144 */
  int64_t tcap = 2 * cap - 1, p = 0;
  while (1) {
    int64_t left = 2 * p + 1;
    if (left >= tcap) break;
    if (v <= tree[left]) p = left; else { v -= tree[left]; p = left + 1; }
  }
  if (prio_out) *prio_out = tree[p];
  return p;
}
double orc_tree_max_leaf(const double* tree, int64_t cap) {                               /* :151-152 */
  double m = tree[cap - 1];
  for (int64_t i = 1; i < cap; ++i) if (tree[cap - 1 + i] > m) m = tree[cap - 1 + i];
  return m;
}
/* S4 — (|td| + eps)^alpha, optional clip (variant B). */
void orc_per_priorities(const float* td, int B, double alpha, double eps, double clip, double* out) {
  for (int i = 0; i < B; ++i) {
    /* :259 `(np.abs(td_errors) + 0.01) ** self.alpha` on a float32 array: float32 throughout
     * (NumPy >= 2 weak python scalars); x**a restated as exp(a*log x) in the reproducible f32 math */
    float e = fabsf(td[i]) + (float)eps;
    if (clip > 0.0 && e > (float)clip) e = (float)clip;
    out[i] = (double)orc_expf((float)alpha * orc_logf(e));
  }
}
/* S3 — stratified sample + IS weights (:220-241; variant B ddqn_per_cartpole.py:119-138). */
void orc_per_sample(const double* tree, int64_t cap, const double* u, uint64_t seed, uint64_t counter, int B,
                    int64_t size, double beta, int variant_b, int32_t* idx_out, double* prio_out, float* w_out) {
  double total = tree[0], segment = total / (double)B;
  double* w64 = (double*)malloc(sizeof(double) * (size_t)B);
  float mx32 = 0.0f; double mx64 = 0.0;
  for (int i = 0; i < B; ++i) {
    double a = segment * (double)i, b = segment * (double)(i + 1), ui;
    if (u) ui = u[i];
    else {
      uint32_t r[4];
      orc_philox(seed, (uint32_t)i, 1u, (uint32_t)counter, RNG_REPLAY | (uint32_t)((counter >> 32) & 0x0FFFFFFFu), r);
      ui = u01d(r[0], r[1]);
    }
    double v = a + (b - a) * ui, pr;
    int64_t p = orc_tree_get_index(tree, cap, v, &pr);
    idx_out[i] = (int32_t)(variant_b ? p : p - cap + 1);
    if (prio_out) prio_out[i] = pr;
    double wd = pow((double)size * (pr / total), -beta);
    if (variant_b) { w64[i] = wd; if (wd > mx64) mx64 = wd; }
    else { w_out[i] = (float)wd; if (w_out[i] > mx32) mx32 = w_out[i]; }
  }
  for (int i = 0; i < B; ++i) w_out[i] = variant_b ? (float)(w64[i] / mx64) : w_out[i] / mx32;
  free(w64);
}

/* ============================================================= n-step ===== */
/* S2 — store_transition + _get_n_step_transition, rainbow_dqn_cartpole.py:179-218, for N
 * independent env windows (window arrays [n][N][...]); emits into ring rows (cursor+e)%cap. */
int orc_nstep_push(float* w_state, int32_t* w_action, float* w_reward, float* w_next, uint8_t* w_terminal,
                   uint8_t* w_done, int n_steps, int64_t pushes, int N, int D, double gamma, const float* obs,
                   const int32_t* action, const float* reward, const float* next_obs, const uint8_t* terminal,
                   const uint8_t* done, float* r_state, uint32_t* r_action, float* r_reward, float* r_next,
                   uint8_t* r_flag, int64_t cap, int64_t cursor) {
  int slot = (int)(pushes % n_steps), emit = pushes + 1 >= n_steps;
  for (int e = 0; e < N; ++e) {
    size_t so = (size_t)slot * N + e;
    memcpy(w_state + so * D, obs + (size_t)e * D, sizeof(float) * D);
    memcpy(w_next + so * D, next_obs + (size_t)e * D, sizeof(float) * D);
    w_action[so] = action[e]; w_reward[so] = reward[e]; w_terminal[so] = terminal[e]; w_done[so] = done[e];
    if (!emit) continue;
    int oldest = (slot + 1) % n_steps, src = slot;
    double R = 0.0;
    for (int i = n_steps - 1; i >= 0; --i) {                       /* :211-216 */
      int s = (oldest + i) % n_steps;
      size_t o = (size_t)s * N + e;
      double d = w_done[o] ? 1.0 : 0.0;
      R = (double)w_reward[o] + gamma * (1.0 - d) * R;
      if (w_done[o]) src = s;
    }
    int64_t row = (cursor + e) % cap;
    size_t oo = (size_t)oldest * N + e, ss = (size_t)src * N + e;
    memcpy(r_state + row * D, w_state + oo * D, sizeof(float) * D);
    memcpy(r_next + row * D, w_next + ss * D, sizeof(float) * D);
    r_action[row] = (uint32_t)w_action[oo]; r_reward[row] = (float)R; r_flag[row] = w_terminal[ss];
  }
  return emit;
}

/* random.sample / np.random.choice(replace=False): the first B elements of a keyed permutation of [0, size)
 * (the bijection of orc_permutation, gymrl_oracle.c, with the replay tag folded into the key). */
static uint32_t replay_feistel(uint32_t x, int a, int b, uint64_t seed, uint64_t counter) {
  const uint32_t mask_lo = (1u << a) - 1u, mask_hi = (1u << b) - 1u;
  uint32_t lo = x & mask_lo, hi = x >> a, o[4];
  const uint32_t c2 = (uint32_t)counter, c3 = 0x60000000u | ((uint32_t)(counter >> 32) & 0x0FFFFFFFu);
  for (uint32_t r = 0; r < 6; ++r) {
    if ((r & 1u) == 0u) { orc_philox(seed, hi, r, c2, c3, o); lo ^= o[0] & mask_lo; }
    else                { orc_philox(seed, lo, r, c2, c3, o); hi ^= o[0] & mask_hi; }
  }
  return (hi << a) | lo;
}
void orc_uniform_indices(uint64_t seed, uint64_t counter, int64_t size, int B, int32_t* idx) {
  int bits = 2;
  while (((int64_t)1 << bits) < size) ++bits;
  const int a = bits / 2, bb = bits - a;
  const uint64_t key = seed ^ 0x5265706C61794944ull;
  for (int b = 0; b < B; ++b) {
    uint32_t x = replay_feistel((uint32_t)b, a, bb, key, counter);
    while (x >= (uint32_t)size) x = replay_feistel(x, a, bb, key, counter);
    idx[b] = (int32_t)x;
  }
}

/* ========================================================== NoisyLinear === */
static float scale_noise(float x) { float s = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); return s * sqrtf(fabsf(x)); }
static float box_muller(uint64_t seed, uint64_t counter, uint32_t stream, uint32_t i) {
  uint32_t r[4];
  orc_philox(seed, i, stream, (uint32_t)counter, RNG_NOISE | (uint32_t)((counter >> 32) & 0x0FFFFFFFu), r);
  float u1 = u01f_open0(r[0]), u2 = u01f(r[1]), s, c;
  orc_sincosf(6.28318530717958647692f * u2, &s, &c);
  return sqrtf(-2.0f * orc_logf(u1)) * c;
}
/* R1 — scale_noise + reset_noise, rainbow_dqn_cartpole.py:77-87. */
void orc_noisy_noise(const float* eps_in, const float* eps_out, uint64_t seed, uint64_t counter, int nin, int nout,
                     float* w_eps, float* b_eps) {
  for (int j = 0; j < nout; ++j) {
    float fj = scale_noise(eps_out ? eps_out[j] : box_muller(seed, counter, 1u, (uint32_t)j));
    b_eps[j] = fj;
    for (int i = 0; i < nin; ++i)
      w_eps[(size_t)j * nin + i] = fj * scale_noise(eps_in ? eps_in[i] : box_muller(seed, counter, 0u, (uint32_t)i));
  }
}

/* D3 — select_action, dqn_cartpole.py:117-133. */
void orc_epsilon_greedy(const float* q, const float* u, uint64_t seed, uint64_t counter, int64_t env_id0, int n,
                        int A, float epsilon, int32_t* act) {
  for (int i = 0; i < n; ++i) {
    float u0, u1;
    if (u) { u0 = u[2 * i]; u1 = u[2 * i + 1]; }
    else {
      uint64_t env = (uint64_t)(env_id0 + i); uint32_t r[4];
      orc_philox(seed, (uint32_t)env, (uint32_t)(env >> 32), (uint32_t)counter,
                 RNG_POLICY | 0x08000000u | (uint32_t)((counter >> 32) & 0x07FFFFFFu), r);
      u0 = u01f(r[0]); u1 = u01f(r[1]);
    }
    int a = 0;
    if (u0 < epsilon) { a = (int)(u1 * (float)A); if (a >= A) a = A - 1; }
    else { for (int k = 1; k < A; ++k) if (q[(size_t)i * A + k] > q[(size_t)i * A + a]) a = k; }
    act[i] = a;
  }
}

/* D4 / R4 — dqn_cartpole.py:157-161, rainbow_dqn_cartpole.py:319-338. */
void orc_dqn_td_loss(const float* q, const float* qn_online, const float* qn_target, const int32_t* act,
                     const float* rew, const float* flag, const float* w, int B, int A, double gamma_n,
                     float* td_out, float* dq_out, double* loss_sum) {
  float g = (float)gamma_n, invB = 1.0f / (float)B;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* sel = qn_online ? qn_online + (size_t)b * A : qn_target + (size_t)b * A;
    int astar = 0;
    for (int k = 1; k < A; ++k) if (sel[k] > sel[astar]) astar = k;
    float y = rew[b] + g * qn_target[(size_t)b * A + astar] * (1.0f - flag[b]);
    float td = q[(size_t)b * A + act[b]] - y, wb = w ? w[b] : 1.0f;
    td_out[b] = td;
    for (int k = 0; k < A; ++k) dq_out[(size_t)b * A + k] = (k == act[b]) ? (2.0f * td) * wb * invB : 0.0f;
    acc += (double)((td * td) * wb);
  }
  if (loss_sum) loss_sum[0] += acc;
}

/* ================================================================ SAC ===== */
/* A1 — Actor.sample, sac_pendulum.py:76-87. */
void orc_sac_sample_fwd(const float* mean, const float* log_std, const float* eps, int B, int A, float bound,
                        float* action, float* logp) {
  const float c = 0.91893853320467274178f;
  for (int b = 0; b < B; ++b) {
    float lp = 0.0f;
    for (int j = 0; j < A; ++j) {
      size_t o = (size_t)b * A + j;
      float mu = mean[o], std = orc_expf(log_std[o]);
      float x = mu + std * eps[o];
      float t = orc_tanhf(x);
      action[o] = t * bound;
      float var = std * std, log_scale = orc_logf(std);
      float l = -((x - mu) * (x - mu)) / (2.0f * var) - log_scale - c;
      l -= orc_logf(bound * (1.0f - t * t) + 1e-6f);
      lp += l;
    }
    logp[b] = lp;
  }
}
void orc_sac_sample_bwd(const float* mean, const float* log_std, const float* eps, const float* d_action,
                        const float* d_logp, int B, int A, float bound, float* d_mean, float* d_log_std) {
  for (int b = 0; b < B; ++b) {
    float gl = d_logp ? d_logp[b] : 0.0f;
    for (int j = 0; j < A; ++j) {
      size_t o = (size_t)b * A + j;
      float mu = mean[o], std = orc_expf(log_std[o]), e = eps[o];
      float x = mu + std * e, t = orc_tanhf(x), omt = 1.0f - t * t;
      float ga = d_action ? d_action[o] : 0.0f;
      float dx = ga * bound * omt + gl * (2.0f * t * bound * omt / (bound * omt + 1e-6f));
      d_mean[o] = dx;
      d_log_std[o] = dx * (std * e) - gl;
    }
  }
}
/* A4 — SACTrainer.update pieces, sac_pendulum.py:233-263. */
void orc_sac_target(const float* rew, const float* done, const float* q1n, const float* q2n, const float* logp_n,
                    const double* log_alpha, int B, double gamma, float* y) {
  float alpha = (float)exp(log_alpha[0]), g = (float)gamma;
  for (int b = 0; b < B; ++b) {
    float tq = fminf(q1n[b], q2n[b]) - alpha * logp_n[b];
    y[b] = rew[b] + g * (1.0f - done[b]) * tq;
  }
}
void orc_sac_critic_loss(const float* q1, const float* q2, const float* y, int B, float* dq1, float* dq2, double* sums) {
  float invB = 1.0f / (float)B; double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    float e1 = q1[b] - y[b], e2 = q2[b] - y[b];
    dq1[b] = 2.0f * e1 * invB; dq2[b] = 2.0f * e2 * invB;
    acc += (double)(e1 * e1) + (double)(e2 * e2);
  }
  sums[0] += acc;
}
void orc_sac_actor_loss(const float* logp, const float* q1, const float* q2, const double* log_alpha, int B,
                        double target_entropy, float* dlogp, float* dq1, float* dq2, double* sums) {
  float invB = 1.0f / (float)B, alpha = (float)exp(log_alpha[0]), te = (float)target_entropy;
  double a0 = 0.0, a1 = 0.0;
  for (int b = 0; b < B; ++b) {
    float a = q1[b], c = q2[b];
    float w1 = a < c ? 1.0f : (a == c ? 0.5f : 0.0f);
    dlogp[b] = alpha * invB; dq1[b] = -w1 * invB; dq2[b] = -(1.0f - w1) * invB;
    a0 += (double)(alpha * logp[b] - fminf(a, c));
    a1 += (double)(logp[b] + te);
  }
  sums[1] += a0; sums[2] += a1;
}
void orc_sac_alpha_step(double* log_alpha, double* m, double* v, const double* sums, int B, double lr, double beta1,
                        double beta2, double eps, int64_t step, double* loss_out) {
  double mean_term = sums[2] / (double)B;
  if (loss_out) loss_out[0] = -(log_alpha[0] * mean_term);
  double g = -mean_term;
  double bc1 = 1.0 - pow(beta1, (double)step), bc2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
  m[0] = m[0] + (g - m[0]) * (1.0 - beta1);
  v[0] = v[0] * beta2 + (1.0 - beta2) * g * g;
  double denom = sqrt(v[0]) / bc2_sqrt + eps;
  log_alpha[0] = log_alpha[0] - (lr / bc1) * (m[0] / denom);
}

/* ====================================================== normalisation ===== */
/* N1/N2 — RunningMeanStd.update + Normalization.__call__, utils/normalization.py:12-35.
 * stats = (n, unused, mean[D], S[D], std[D]); rows consumed in order. */
void orc_running_norm(const float* x, int N, int D, double* stats, int update, float* y) {
  for (int i = 0; i < N; ++i) {
    if (update) stats[0] += 1.0;
    double n = stats[0];
    for (int k = 0; k < D; ++k) {
      float xv = x[(size_t)i * D + k];
      float mean = (float)stats[2 + k];
      double S = stats[2 + D + k], std = stats[2 + 2 * D + k];
      if (update) {
        if (n == 1.0) { mean = xv; std = (double)xv; }
        else {
          float old_mean = mean;
          mean = old_mean + (xv - old_mean) / (float)n;
          S = S + (double)((xv - old_mean) * (xv - mean));
          std = sqrt(S / n);
        }
        stats[2 + k] = (double)mean; stats[2 + D + k] = S; stats[2 + 2 * D + k] = std;
      }
      y[(size_t)i * D + k] = (float)((double)(xv - mean) / (std + 1e-8));
    }
  }
}
/* N3 — RewardScaling, utils/normalization.py:38-52 (per-env R, shared statistics). */
void orc_reward_scaling(const float* r, const uint8_t* done, int N, double gamma, double* R, double* stats, float* y) {
  for (int i = 0; i < N; ++i) {
    double rv = (double)r[i];
    R[i] = gamma * R[i] + rv;
    float xv = (float)R[i];
    stats[0] += 1.0;
    double n = stats[0];
    float mean = (float)stats[2];
    if (n == 1.0) { mean = xv; stats[4] = (double)xv; }
    else {
      float old_mean = mean;
      mean = old_mean + (xv - old_mean) / (float)n;
      stats[3] = stats[3] + (double)((xv - old_mean) * (xv - mean));
      stats[4] = sqrt(stats[3] / n);
    }
    stats[2] = (double)mean;
    y[i] = (float)(rv / (stats[4] + 1e-8));
    if (done && done[i]) R[i] = 0.0;
  }
}

/* ============================================================ TD3 / DDPG === */
/* gymrl_noisy_action — ddpg_pendulum.py:143-147, td3_pendulum.py:164-168 (mode 0, numpy float64) and
 * td3_pendulum.py:191-196 (mode 1, torch float32 target-policy smoothing). */
void orc_noisy_action(const float* mu, const double* eps, uint64_t seed, uint64_t counter, int64_t n, int mode,
                      double std, double noise_clip, double bound, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    double e = eps ? eps[i] : (double)box_muller(seed, counter, 2u, (uint32_t)i);
    if (mode == 0) {
      double a = (double)mu[i] + e * std;
      a = a < -(double)(float)bound ? -(double)(float)bound : (a > (double)(float)bound ? (double)(float)bound : a);
      out[i] = (float)a;
    } else {
      float nz = (float)e * (float)std, c = (float)noise_clip, b = (float)bound;
      nz = fminf(fmaxf(nz, -c), c);
      out[i] = fminf(fmaxf(mu[i] + nz, -b), b);
    }
  }
}

/* gymrl_mse_loss — F.mse_loss(current_q, target_q), ddpg_pendulum.py:178-179. */
void orc_mse_loss(const float* q, const float* y, int B, float* dq, double* sum) {
  double acc = 0.0; float invB = 1.0f / (float)B;
  for (int b = 0; b < B; ++b) { float e = q[b] - y[b]; dq[b] = 2.0f * e * invB; acc += (double)(e * e); }
  *sum += acc;
}

/* gymrl_neg_mean_loss — -critic(states, actor(states)).mean(), ddpg_pendulum.py:185, td3_pendulum.py:213. */
void orc_neg_mean_loss(const float* q, int B, float* dq, double* sum) {
  double acc = 0.0; float g = -1.0f / (float)B;
  for (int b = 0; b < B; ++b) { dq[b] = g; acc += (double)q[b]; }
  *sum += acc;
}

/* ========================================================= discrete SAC === */
/* sac_cartpole.py:171-181 */
void orc_dsac_target(const float* probs_n, const float* q1n, const float* q2n, const float* rew, const float* done,
                     const float* log_alpha, int B, int A, float gamma, float* y) {
  float alpha = orc_expf(log_alpha[0]);
  for (int b = 0; b < B; ++b) {
    float ent = 0.0f, minq = 0.0f;
    for (int k = 0; k < A; ++k) {
      float p = probs_n[(size_t)b * A + k];
      ent += p * orc_logf(p + 1e-8f);
      minq += p * fminf(q1n[(size_t)b * A + k], q2n[(size_t)b * A + k]);
    }
    float nv = minq + alpha * (-ent);
    y[b] = rew[b] + gamma * (1.0f - done[b]) * nv;
  }
}

/* :183-186 */
void orc_dsac_critic_loss(const float* q1, const float* q2, const int32_t* act, const float* y, int B, int A, float* dq1,
                          float* dq2, double* sums) {
  float invB = 1.0f / (float)B;
  for (int b = 0; b < B; ++b) {
    int a = act[b];
    float e1 = q1[(size_t)b * A + a] - y[b], e2 = q2[(size_t)b * A + a] - y[b];
    for (int k = 0; k < A; ++k) {
      dq1[(size_t)b * A + k] = k == a ? 2.0f * e1 * invB : 0.0f;
      dq2[(size_t)b * A + k] = k == a ? 2.0f * e2 * invB : 0.0f;
    }
    sums[0] += (double)(e1 * e1); sums[1] += (double)(e2 * e2);
  }
}

/* :196-203 */
void orc_dsac_actor_loss(const float* probs, const float* q1, const float* q2, const float* log_alpha, int B, int A,
                         float* dprobs, double* sums) {
  float invB = 1.0f / (float)B, alpha = orc_expf(log_alpha[0]);
  for (int b = 0; b < B; ++b) {
    float ent = 0.0f, minq = 0.0f;
    for (int k = 0; k < A; ++k) {
      float p = probs[(size_t)b * A + k], lp = orc_logf(p + 1e-8f);
      float m = fminf(q1[(size_t)b * A + k], q2[(size_t)b * A + k]);
      ent += p * lp; minq += p * m;
      dprobs[(size_t)b * A + k] = (alpha * (lp + p / (p + 1e-8f)) - m) * invB;
    }
    ent = -ent;
    sums[0] += (double)(-alpha * ent - minq); sums[1] += (double)ent;
  }
}

/* :209-215 — float32 Adam step on log_alpha */
void orc_dsac_alpha_step(float* log_alpha, float* m, float* v, const double* sums, int B, float target_entropy, float lr,
                         float b1, float b2, float eps, int64_t step, double* loss_out) {
  float alpha = orc_expf(log_alpha[0]);
  float mean_gap = (float)(sums[1] / (double)B) - target_entropy;
  if (loss_out) loss_out[0] = (double)(alpha * mean_gap);
  float g = alpha * mean_gap;
  float mm = b1 * m[0] + (1.0f - b1) * g, vv = b2 * v[0] + (1.0f - b2) * g * g;
  m[0] = mm; v[0] = vv;
  double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float denom = (float)(sqrt((double)vv) / sqrt(bc2)) + eps;
  log_alpha[0] = log_alpha[0] - step_size * (mm / denom);
}
