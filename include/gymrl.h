/*
 * gymrl.h — C-ABI of libgymrl_hip.so: the MI355X (gfx950) vectorised-rollout +
 * PPO/DQN/SAC learner hot path that sits behind the Config / Trainer surface of
 * Starlight0798/gymRL's algorithms/<algo>_<env>.py scripts.
 *
 * The reference has no FFI of its own (pure Python); every entry point below
 * cites the reference function (file:line under the gymRL tree) whose arithmetic
 * it replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every data pointer is a DEVICE pointer (HBM) unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - return 0 on success, -EINVAL (-22) on a bad argument, -(hipError_t) - 1000
 *     when a launch fails.  Nothing allocates, nothing synchronises, nothing
 *     copies to the host: the caller owns every buffer.  Safe to capture in a
 *     hipGraph.
 *   - layouts are time-major SoA slabs: x[T][N] (t outer, env inner) so that a
 *     wavefront's 64 lanes (= 64 env instances) touch 64 consecutive words.
 */
#ifndef GYMRL_H
#define GYMRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GYMRL_ABI_VERSION 3   /* 2: round-2 signature changes (adam_step, per_*, replay_append, nstep_push, gae_decoupled, rollout descriptors); gymrl_gemm_config left the product library
                               * 3: gymrl_rollout_lunar_args grew `gae_carry` (round 6), gymrl_gae variant 3 */

/* ---------------------------------------------------------------- misc --- */
int gymrl_abi_version(void);
/* 1 when a gfx950 device is visible to the HIP runtime, else 0. */
int gymrl_device_ok(void);

/* ------------------------------------------------------------- env kinds -- */
enum {
  GYMRL_ENV_CARTPOLE    = 0, /* CartPole-v1   obs 4, 2 discrete actions   */
  GYMRL_ENV_PENDULUM    = 1, /* Pendulum-v1   obs 3, 1 continuous in [-2,2] */
  GYMRL_ENV_LUNARLANDER = 2  /* LunarLander-v3 obs 8, 4 discrete actions  */
};

/*
 * Batched env stepper.  Replaces gym.make / env.reset / env.step as used by
 * ppo_lunarlander.py:160,200,211,222  dqn_cartpole.py:94,176,181
 * rainbow_dqn_cartpole.py:270,369,373  sac_pendulum.py:154,275,281
 * utils/runner.py:53,111,123.
 * One env instance per lane; state is a caller-owned SoA byte buffer of
 * gymrl_env_state_bytes(kind, n_envs) bytes (256-B aligned base required).
 * Classic envs: consecutive fields [n_envs], each field's byte size rounded up to a multiple of 256 —
 *   CartPole-v1  f64 x, f64 x_dot, f64 theta, f64 theta_dot, f64 ep_ret, i32 ep_len, u32 episode
 *   Pendulum-v1  f64 theta, f64 theta_dot, f64 ep_ret, i32 ep_len, u32 episode
 * (float64 like gymnasium's own state; tests/test_classic_micro_gpu.py writes states through this layout, and a
 * checkpoint of a running vector is a copy of the buffer).  LunarLander: 144 dwords per env, word-major.
 */
int    gymrl_env_obs_dim(int kind);
int    gymrl_env_act_dim(int kind);     /* #discrete actions, or continuous dim */
int    gymrl_env_is_discrete(int kind);
int    gymrl_env_max_steps(int kind);   /* TimeLimit: 500 / 200 / 1000 */
size_t gymrl_env_state_bytes(int kind, int n_envs);

/* (Re)initialise every env: episode counter := 0, draws from the counter-based
 * Philox stream keyed by (seed, env_id0 + lane, episode).  obs_out [N, obs_dim]. */
int gymrl_env_reset(int kind, void* state, int n_envs, uint64_t seed,
                    int64_t env_id0, float* obs_out, void* stream);

/* One vector step with auto-reset.
 *   action        i32[N] (discrete) or f32[N, act_dim] (continuous)
 *   obs_out       f32[N, obs]  observation the policy sees next (post-reset
 *                 where done — ppo_lunarlander.py:220-223)
 *   term_obs_out  f32[N, obs] or NULL: pre-reset observation (what off-policy
 *                 buffers store as next_state — dqn_cartpole.py:183)
 *   rew_out       f32[N]
 *   terminated_out / truncated_out  u8[N]  (rainbow_dqn_cartpole.py:376 needs both)
 *   done_out      u8[N] or NULL  = terminated | truncated (ppo_lunarlander.py:212)
 *   ep_ret_out    f32[N] or NULL: finished episode's return where done, else unchanged
 *   ep_len_out    i32[N] or NULL: finished episode's length where done
 *   ep_stats      f64[3] or NULL: += (#finished episodes, sum of returns, sum of lengths)
 */
int gymrl_env_step(int kind, void* state, int n_envs, uint64_t seed, int64_t env_id0,
                   const void* action, float* obs_out, float* term_obs_out,
                   float* rew_out, uint8_t* terminated_out, uint8_t* truncated_out,
                   uint8_t* done_out, float* ep_ret_out, int32_t* ep_len_out,
                   double* ep_stats, void* stream);

/* Episodes a trainer abandons at its own step cap below the env's TimeLimit — `for step in range(cfg.max_steps)`
 * of dqn_cartpole.py:178, sac_pendulum.py:278 (no done flag is stored; the next loop turn calls env.reset()).
 * Call after gymrl_env_step: every env whose running episode has reached `cap` steps starts its next episode;
 * obs_inout [N, obs] rows of those envs become the reset observation, flag_inout u8[N] (or NULL) is OR-ed with 1
 * for them, ep_ret_out / ep_len_out / ep_stats receive the abandoned episode as gymrl_env_step would for a
 * finished one.  CartPole-v1 and Pendulum-v1 (no reference off-policy script runs LunarLander: -EINVAL). */
int gymrl_env_abandon(int kind, void* state, int n_envs, uint64_t seed, int64_t env_id0, int cap,
                      float* obs_inout, uint8_t* flag_inout, float* ep_ret_out, int32_t* ep_len_out,
                      double* ep_stats, void* stream);

/* Optional latency hiding for expensive resets (LunarLander's reset() ends with a full
 * physics step): prepares, for every env that lacks one, the post-reset world of its NEXT
 * episode in a spare slot of `state`; gymrl_env_step then swaps it in when the episode ends
 * instead of running the reset inside the step.  Results are bit-identical with or without
 * it.  Intended for a side stream (it may overlap later gymrl_env_step calls on the same
 * state); a no-op for CartPole / Pendulum. */
int gymrl_env_refill(int kind, void* state, int n_envs, uint64_t seed, int64_t env_id0,
                     void* stream);

/* -------------------------------------------------- categorical policy ---- */
/*
 * Categorical(logits).sample()/log_prob/entropy — ppo_lunarlander.py:92-104.
 * sample == argmax_k(softmax(z)_k / q_k), q ~ Exp(1)  (torch.multinomial's CPU
 * path).  noise_exp f32[N,A] supplies q explicitly (parity
 * mode); when NULL q is drawn in-kernel from Philox(seed, counter, env).
 * deterministic != 0 -> action = argmax(z) (ppo_lunarlander.py:98-99).
 * value_in f32[N] (may be NULL) is copied to value_out so the rollout slab row
 * is written by one kernel.  A <= 8.
 */
/* Optional producer-side fusion of GAE's chunk reduction (see gymrl_gae variant 2): while
 * sampling step t the kernel also composes step t-1 — whose delta needs V_t = value_in —
 * into the affine map of its time chunk.  All pointers are rows of the rollout slab. */
typedef struct {
  const float* rew_prev;      /* f32[N] rewards of step t-1                         */
  const uint8_t* done_prev;   /* u8[N]  done flags of step t-1                      */
  const float* val_prev;      /* f32[N] values of step t-1                          */
  double* running;            /* f64[2][N] map of the chunk being composed (scratch) */
  void* gae_workspace;        /* the workspace later handed to gymrl_gae(variant 2)  */
  int t_prev;                 /* t-1, 0-based                                        */
  int T;                      /* rollout length                                      */
  double gamma, lam;
  double lam2;                /* > 0 with running2 != NULL: decoupled-lambda mode (G3) — `lam` is lam_actor,   */
  double* running2;           /* `lam2` lam_critic; both decay factors are the float64 products gamma * lam;   */
                              /* the critic's chunk maps follow the actor's in the workspace of                */
                              /* gymrl_gae_decoupled(variant 2); running2 f64[2][N] scratch                    */
} gymrl_gae_online;

int gymrl_categorical_sample(const float* logits, const float* value_in,
                             const float* noise_exp, uint64_t seed, uint64_t counter,
                             int64_t env_id0, int n, int n_actions, int deterministic,
                             int32_t* act_out, float* logp_out, float* ent_out,
                             float* value_out, const gymrl_gae_online* online, void* stream);
/* Last step of a rollout: composes step T-1 with val_cur = the bootstrap value V_T. */
int gymrl_gae_online_flush(const gymrl_gae_online* online, const float* val_cur, int N,
                           void* stream);
int gymrl_gae_chunk(void);   /* time-chunk length of the blocked GAE (16) */

/* ------------------------------------------------------------------ GAE --- */
/*
 * G1: PPOTrainer.compute_gae — ppo_lunarlander.py:179-196.
 *   delta_t = r_t + g*V_{t+1}*(1-d_t) - V_t ; A_t = delta_t + g*l*(1-d_t)*A_{t+1}
 *   V_T = next_val ; ret = A + V.   float64 recursion, float32 storage.
 * rew,val f32[T][N]; done u8[T][N]; next_val f32[N]; adv_out, ret_out f32[T][N].
 * moments_out f64[3] or NULL: = (count, sum A, sum A^2) over the float64
 * advantages (feeds the whole-rollout normalisation of ppo_lunarlander.py:236);
 * reduced in a fixed order (no float atomics) so training is reproducible.
 * variant 0 = one lane walks one env sequentially (reference operation order);
 * variant 1 = time-blocked affine scan (roofline variant; needs N % 4 == 0 and
 *             16-B aligned rows, else it falls back to variant 0);
 * variant 2 = variant 1 without its first pass: the per-chunk maps were composed during
 *             the rollout (gymrl_gae_online in gymrl_categorical_sample + _flush), so the
 *             launch reads r, v, d once (17 -> ~19 HBM bytes per transition instead of 29).
 * variant 3 = variant 2 without its carry launch: the persistent rollout ran the carry pass for its own envs
 *             at its tail (gymrl_rollout_lunar_args.gae_carry) — the apply launch (+ the moments' fold) alone;
 *             the carries, and so every advantage, have variant 2's bits.
 * workspace: gymrl_gae_workspace_bytes(T, N) bytes, 256-B aligned; required for
 * variant 1 or when moments_out != NULL.
 */
size_t gymrl_gae_workspace_bytes(int T, int N);
int gymrl_gae(const float* rew, const float* val, const uint8_t* done,
              const float* next_val, int T, int N, double gamma, double lam,
              float* adv_out, float* ret_out, double* moments_out,
              int variant, void* workspace, void* stream);

/* G2: ReplayBuffer_on_policy.compute_advantage — utils/buffer.py:21-35
 * (== ppo_rnn_lunarlander.py:187-204): stored next values, separate dw
 * (terminated) and done masks, float32 recursion.  All arrays [T][N]. */
int gymrl_gae_dw(const float* rew, const float* val, const float* next_val,
                 const uint8_t* done, const uint8_t* dw, int T, int N,
                 double gamma, double lam, float* adv_out, float* vtarget_out,
                 double* moments_out, void* workspace, void* stream);

/* G3: ppo_full compute_advantages — ppo_full_lunarlander.py:507-535: two scans
 * sharing delta (lam_actor, lam_critic); ret = adv_critic + V; adv_actor raw.
 * variant 0 = one lane walks one env sequentially (the reference's operation order);
 * variant 1 = time-blocked affine scan with two (A, b) maps per chunk (same layout rules as gymrl_gae variant 1,
 *             falls back to variant 0 when they do not hold);
 * variant 2 = variant 1 without its first pass: both chunk maps were composed during the rollout
 *             (gymrl_gae_online with lam2 / running2).
 * workspace: gymrl_gae_decoupled_workspace_bytes(T, N) bytes, 256-B aligned (variants 1, 2; NULL: variant 0). */
size_t gymrl_gae_decoupled_workspace_bytes(int T, int N);
int gymrl_gae_decoupled(const float* rew, const float* val, const uint8_t* done,
                        const float* next_val, int T, int N, double gamma,
                        double lam_actor, double lam_critic, float* adv_actor_out,
                        float* ret_out, int variant, void* workspace, void* stream);

/* Whole-rollout advantage normalisation — ppo_lunarlander.py:236 (ddof 0) and
 * utils/buffer.py:33 (ddof 1).  moments f64[3] = (count, sum, sumsq) on device.
 * x <- (x - mean) / (std + eps), f64 arithmetic, f32 storage. */
size_t gymrl_reduce_workspace_bytes(void);
int gymrl_moments(const float* x, int64_t n, double* moments_out, void* workspace,
                  void* stream);
int gymrl_normalize(float* x, int64_t n, const double* moments, int ddof,
                    double eps, void* stream);

/* ------------------------------------------------------------- PPO loss --- */
typedef struct {
  float clip_eps;     /* ppo_lunarlander.py:41  (0.2)  */
  float dual_clip;    /* :42 (3.0)                     */
  float value_coef;   /* :44 (0.5)                     */
  float entropy_coef; /* :43 (0.01)                    */
} gymrl_ppo_cfg;

/*
 * L1+L2: evaluate_actions + clipped/dual-clip surrogate + value + entropy loss,
 * forward AND backward w.r.t. logits/value, plus the five metrics —
 * ppo_lunarlander.py:110-117, 278-300, 309-322.
 *   logits f32[B,A], value f32[B]: network outputs for minibatch rows 0..B-1
 *   idx    i32[B] or NULL: row b reads act/logp_old/adv/ret at idx[b] of the
 *          rollout slab (fused minibatch gather, ppo_lunarlander.py:268-272)
 *   act i32[*], logp_old f32[*], adv f32[*], ret f32[*]
 *   adv_moments f64[3] or NULL: when given, adv is normalised on the fly with
 *          (mean, population std + 1e-8) — ppo_lunarlander.py:236
 *   dlogits_out f32[B,A], dvalue_out f32[B]: d(loss)/d(logits|value), loss =
 *          mean over the B rows (torch autograd tie rules)
 *   metrics_sum f64[5] or NULL: += (sum -obj, vf*sum (v-ret)^2, sum H, #clipped,
 *          sum (logp_old - logp)).  The caller keeps one zeroed row per minibatch
 *          and divides by B on the host after the whole update (one D2H copy
 *          instead of ppo_lunarlander.py:309-322's five .item() per minibatch).
 *          Reduced block partials -> fixed-order sum through `workspace`
 *          (>= gymrl_reduce_workspace_bytes(), required when metrics_sum != NULL).
 *          With metrics_sum == NULL and workspace != NULL the kernel only writes its
 *          block partials f64[gymrl_loss_blocks(B)][5] to `workspace`: give every minibatch
 *          its own slice of one table and reduce the whole update with ONE
 *          gymrl_reduce_rows launch (what the trainers do).
 */
int gymrl_ppo_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                           const int32_t* act, const float* logp_old, const float* adv,
                           const float* ret, const double* adv_moments, int B, int A,
                           const gymrl_ppo_cfg* cfg_host, float* dlogits_out,
                           float* dvalue_out, double* metrics_sum, void* workspace,
                           void* stream);

int gymrl_loss_blocks(int B);
/* out[r][k] = sum over b < blocks_per_row of partials[r][b][k]  (f64, fixed order) */
int gymrl_reduce_rows(const double* partials, int rows, int blocks_per_row, int K,
                      double* out, void* stream);

typedef struct {
  float clip_eps_min;  /* ppo_full_lunarlander.py:34 (0.2)  */
  float clip_eps_max;  /* :35 (0.28)                        */
  float dual_clip;     /* :36 (3.0) ratio clamp upper bound */
  float erc_beta_low;  /* :39 (0.06)                        */
  float erc_beta_high; /* :40 (0.06)                        */
  float entropy_coef;  /* current (annealed) value, :662-666 */
  const float* entropy_coef_dev;   /* the same from device memory (a hipGraph replayed across updates: the coefficient is
                                    * annealed after every update) or NULL */
} gymrl_ppo_full_cfg;

/* L3: ppo_full update_model minibatch loss — ppo_full_lunarlander.py:575-652.
 * ent_old f32[*] = behaviour-policy entropy stored at collection (:488).
 * corr_mul f32[B] or NULL: per-row multiplier of the entropy-ratio mask — the covariance clip :611-616
 * (clip_cov_ratio > 0; dead under the reference default :44): the caller picks the rows (covs window, random
 * subset of clip_cov_ratio of them) and passes 0 for those, 1 elsewhere.
 * metrics_sum f64[9]: += (sum policy term, sum 0.5*corr*(v-ret)^2, sum H*corr,
 * sum clipped*corr, sum (logp_old-logp), #erc-masked, sum logp, sum adv,
 * sum logp*adv) — the last three give covs.mean() (:594-596) per minibatch. */
int gymrl_ppo_full_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                                const int32_t* act, const float* logp_old,
                                const float* ent_old, const float* adv, const float* ret,
                                int B, int A, const gymrl_ppo_full_cfg* cfg_host, const float* corr_mul,
                                float* dlogits_out, float* dvalue_out,
                                double* metrics_sum, void* workspace, void* stream);

/* L4: recurrent-PPO minibatch loss — ppo_lstm_lunarlander.py:716-776.  The L3 terms with
 * (a) every mean a masked mean over the entropy-ratio mask `corr`: sum(x*corr)/sum(corr), 0 when the mask is
 * empty (masked_mean :646-655), and (b) the clipped value loss :763-770
 *   0.5 * masked_mean(max((v - ret)^2, (v_old + clamp(v - v_old, -clip_eps_min, +clip_eps_max) - ret)^2)).
 * val_old f32[*] = values stored at collection.  The RND loss (:775) is a plain mean of squares and stays with
 * the network's autograd.  metrics_sum f64[10]: the nine L3 sums (index 1 = sum 0.5*corr*max(...)) followed by
 * sum(corr); the caller divides sums 0..3 by metrics_sum[9] (or reports 0 when it is 0).
 * corr_mul as in gymrl_ppo_full_loss_fwd_bwd (:747-753).
 * workspace >= gymrl_reduce_workspace_bytes(). */
int gymrl_ppo_rnn_loss_fwd_bwd(const float* logits, const float* value, const int32_t* idx,
                               const int32_t* act, const float* logp_old, const float* ent_old,
                               const float* val_old, const float* adv, const float* ret, int B, int A,
                               const gymrl_ppo_full_cfg* cfg_host, const float* corr_mul, float* dlogits_out,
                               float* dvalue_out, double* metrics_sum, void* workspace, void* stream);

/* Pointwise half of torch.nn.GRU's cell as used by URNN — ppo_lstm_lunarlander.py:449-491 (nn.GRU,
 * batch_first, one layer).  gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh, both f32[B,3H] in PyTorch's gate
 * order (r, z, n); h f32[B,H]:
 *   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h_out = (1 - z)*n + z*h.
 * _bwd recomputes the gates from (gi, gh) and writes dgi, dgh f32[B,3H] and the direct dh f32[B,H]
 * (dh_out * z; the path through gh is the caller's GEMM).  H % 4 == 0, 16-byte aligned pointers. */
int gymrl_gru_cell_fwd(const float* gi, const float* gh, const float* h, int B, int H, float* h_out,
                       void* stream);
int gymrl_gru_cell_bwd(const float* gi, const float* gh, const float* h, const float* dh_out, int B, int H,
                       float* dgi, float* dgh, float* dh, void* stream);

/* RND intrinsic reward — ppo_lstm_lunarlander.py:588-590: rnd[b] = mean_e (predict[b,e] - target[b,e])^2
 * (float32, as numpy's mean of a float32 array), rew_inout[b] += rnd[b] when given, rnd_out[b] = rnd[b]
 * when given.  One wave per row: lane l adds columns l, l+64, ... in order, then a shfl_down tree. */
int gymrl_rnd_reward(const float* predict, const float* target, int B, int E, float* rew_inout,
                     float* rnd_out, void* stream);

/*
 * P6/P7: minibatch staging — ppo_lunarlander.py:238-272 (lists -> tensors, shuffled
 * index slices).  gymrl_pack_rollout writes one 64-B record per transition
 *   packed[i] = { obs[i][0..D) (zero padded to 12) | act bits | logp_old | adv | ret }
 * once per rollout; gymrl_gather_minibatch then fetches ONE random cache line per
 * sample (idx i32[B] = a slice of the epoch's permutation) into contiguous rows
 * obs_out f32[B,D], act_out i32[B], logp_out/adv_out/ret_out f32[B].  D <= 12.
 */
/* P6: the epoch shuffle — ppo_lunarlander.py:262 `np.random.shuffle(indices)`.  perm_out i32[M] = a keyed
 * bijection of [0, M) evaluated per element: 6 alternating Feistel rounds on ceil(log2 M) bits with a
 * Philox4x32-10 round function keyed by (seed, counter), cycle-walked into range.  One streaming write of 4M
 * bytes instead of torch.randperm's key sort (3.3 ms per epoch at M = 2^23).  Deterministic in (seed, counter, M);
 * any uniform-looking permutation serves the algorithm, and parity runs pass the reference's own order instead. */
int gymrl_permutation(uint64_t seed, uint64_t counter, int64_t M, int32_t* perm_out, void* stream);
int gymrl_pack_rollout(const float* obs, const int32_t* act, const float* logp,
                       const float* adv, const float* ret, int64_t M, int obs_dim,
                       float* packed, void* stream);
int gymrl_gather_minibatch(const float* packed, const int32_t* idx, int B, int obs_dim,
                           float* obs_out, int32_t* act_out, float* logp_out,
                           float* adv_out, float* ret_out, void* stream);
/* out [B, row_floats] = src[idx[b], :] — a minibatch of rows by index (ppo_full_lunarlander.py:574 `states[mb_idx]`);
 * row_floats a multiple of 4, src and out 16-byte aligned. */
int gymrl_gather_rows(const float* src, const int32_t* idx, int B, int row_floats, float* out, void* stream);

/* ------------------------------------------------------------ optimiser --- */
/*
 * O1 / D4 / R4: clip_grad_norm_ + Adam on ONE flat fp32 parameter buffer —
 * ppo_lunarlander.py:169,302-307; dqn_cartpole.py:163-166 (clamp_abs = 1);
 * rainbow_dqn_cartpole.py:343-345; sac_pendulum.py:244-255.
 *   gymrl_sqnorm:    sqnorm_out f64[1] (device) = sum (g*grad_scale)^2, fixed-order
 *                    reduction; workspace >= gymrl_reduce_workspace_bytes().  The order is part of the
 *                    contract (oracle/gymrl_oracle.c orc_sqnorm restates it; the optimiser step is pinned bit for
 *                    bit): nb = clamp(ceil(n/4096), 1, 1024) workgroups of 256; thread t of workgroup b adds the
 *                    float4 groups b*256+t, +nb*256, ... as ((a^2+b^2)+c^2)+d^2 in f64; the n%4 tail goes to
 *                    threads 0..2 of workgroup 0; waves fold by halves (32, 16, .. 1), the 4 wave sums add in
 *                    order; the final pass gives thread t partials t, t+256, ... and folds 256 sums by halves.
 *   gymrl_adam_step: scale = max_grad_norm>0 ? min(1, max_norm/(sqrt(sqnorm)+1e-6)) : 1
 *                    g' = clamp_abs>0 ? clamp(g*grad_scale, +-clamp_abs) : g*grad_scale*scale
 *                    m,v,p updated as torch.optim.Adam (no amsgrad, no decay);
 *                    g is zeroed (zero_grad fused) when zero_grad != 0.
 *   lr_dev f32[1] or NULL (then lr_host is used): device-resident lr lets LR
 *   annealing change the rate without re-capturing a graph.
 *   grad_scale multiplies g first (1/world_size after an all-reduce SUM).
 *   bias_dev f32[4] or NULL: device-resident {lr/(1-b1^t), 1/(1-b1^t), sqrt(1-b2^t), 0} — the only
 *   step-dependent scalars of the update.  With it (and `step` ignored) a hipGraph captured around the
 *   optimiser step replays unchanged; the host refreshes the block before each replay with
 *   gymrl_adam_bias (host arithmetic, identical to the eager path's) + gymrl_store_scalars.
 *   polyak_target f32[n] or NULL: theta' <- tau*theta + (1-tau)*theta' (gymrl_soft_update's expression) on the
 *   parameters this launch has just written — optimiser step + soft target update in one pass.
 *   gymrl_store_scalars: copies nbytes <= 3840 (multiple of 4) of host scalars into device memory as ONE
 *   launch whose payload is the kernel argument (stream-ordered, nothing to fence, not capturable state).
 *
 * Per-step scalars on the device.  Entry points whose arguments change every vector step — ring cursors, Philox
 * draw counters, the PER exponent — take a trailing `*_dev` / `dev` pointer (NULL on the plain path): when given,
 * the kernel reads those values from DEVICE memory instead of its host arguments, so the launch can be recorded
 * into a hipGraph and replayed with new values that the host stages beforehand with ONE gymrl_store_scalars for a
 * whole chunk of vector steps (gymrl_amd/graphs.py StepChunk: acting, env step, ring append, index draw and update
 * of 16 vector steps = one scalar store + one graph launch).  Layouts (8-byte aligned):
 *   gymrl_replay_append   cursor_dev     int64[1]  {cursor}
 *   gymrl_uniform_indices dev            {uint64 counter; int64 size}
 *   gymrl_nstep_push      dev            int64[2]  {pushes, cursor}   (the return value follows the HOST pushes)
 *   gymrl_per_update      idx_start_dev  int64[1]  {idx_start}        (idx == NULL only)
 *   gymrl_per_sample      dev            {uint64 counter; int64 size; double beta}
 *   gymrl_noisy_noise     counter_dev    uint64[1] {counter}
 */
int gymrl_sqnorm(const float* g, int64_t n, float grad_scale, double* sqnorm_out,
                 void* workspace, void* stream);
int gymrl_adam_step(float* p, float* g, float* m, float* v, int64_t n,
                    double lr_host, const float* lr_dev, double beta1, double beta2,
                    double eps, int64_t step, const float* bias_dev, float grad_scale,
                    float max_grad_norm, const double* sqnorm, float clamp_abs, int zero_grad,
                    float* polyak_target, double tau, void* stream);
/*
 * clip_grad_norm_ + Adam as TWO launches instead of gymrl_sqnorm's two + gymrl_adam_step's one: the squared norm's first level,
 * then the Adam kernel, every workgroup of which folds the block partials itself exactly as gymrl_sqnorm's second launch does.
 * Same sums in the same order, same bits (tests/test_hip_parity.py).  max_grad_norm > 0; sqnorm_out f64[1] or NULL (the squared
 * norm, for logging); workspace: gymrl_reduce_workspace_bytes().  Other arguments as gymrl_adam_step.
 */
int gymrl_clip_adam_step(float* p, float* g, float* m, float* v, int64_t n, double lr_host, const float* lr_dev, double beta1,
                         double beta2, double eps, int64_t step, const float* bias_dev, float grad_scale, float max_grad_norm,
                         double* sqnorm_out, float clamp_abs, int zero_grad, float* polyak_target, double tau, void* workspace,
                         void* stream);

int gymrl_adam_bias(double lr, double beta1, double beta2, int64_t step, float* out_host4);
int gymrl_store_scalars(void* dst_dev, const void* src_host, int nbytes, void* stream);

/* R4 / A4: theta' <- tau*theta + (1-tau)*theta' on flat buffers —
 * rainbow_dqn_cartpole.py:347-352, sac_pendulum.py:194-199. */
int gymrl_soft_update(float* target, const float* source, int64_t n, double tau,
                      void* stream);

/* ============================================================ MLP forward == */
/*
 * P1 / D1 / A1 inference path: ActorCritic.forward ppo_lunarlander.py:86-90 (as called by
 * get_action :92-104 and get_value :106-108 once per env step), QNetwork.forward
 * dqn_cartpole.py:62-65 and Actor.forward sac_pendulum.py:66-74 in select_action.  The
 * reference runs one torch Linear + activation call per layer on a B = 1 batch; here the
 * whole chain of Linear(+Tanh|ReLU) layers runs in ONE launch on [n_rows, in_dim] rows, 16
 * rows per workgroup, activations in LDS, f32 MFMA (v_mfma_f32_16x16x4_f32: exact f32).
 *
 * A network is a list of <= GYMRL_MLP_MAX_STAGES stages  y = act(x W^T + b):
 *   W = the PACKED image of the f32[out_dim, in_dim] row-major weight (torch
 *   nn.Linear.weight) that gymrl_mlp_pack writes — gymrl_mlp_packed_floats(out_dim, in_dim)
 *   floats, P[tile][kblock][lane][c] = W[16 tile + (lane & 15)][16 kblock + 4 (lane >> 4) + c],
 *   zero padded to 16-row tiles and to K rounded up to 64; repack after every parameter
 *   update (one tiny launch per layer).  b f32[out_dim] or NULL (read in place);
 *   src = -1 reads the network input x, 0..2 reads LDS buffer `src` (written by an earlier
 *   stage); dst = 0..2 writes LDS buffer `dst` (out_dim <= GYMRL_MLP_MAX_WIDTH), dst = -1
 *   writes out[row*out_stride + col] in HBM.  in_dim <= GYMRL_MLP_MAX_WIDTH (network input
 *   <= GYMRL_MLP_MAX_INPUT).  Two heads on one trunk are two stages reading the same src.
 * Summation order (bit-reproducible on the CPU, see csrc/mlp.hip): per output element
 *   acc = 0; for kb in 0,16,.. < roundup(in_dim,64): for j in 0..3: for q in 0..3:
 *     acc = fmaf(x[kb+4q+j], W[n][kb+4q+j], acc)   (k >= in_dim contribute fmaf(0,0,acc));
 *   y = act(acc + b[n]); tanh = the deterministic
 *   polynomial/exp kernel of csrc/gymrl_device.hpp.
 */
#define GYMRL_MLP_MAX_STAGES 8
#define GYMRL_MLP_MAX_WIDTH 256
#define GYMRL_MLP_MAX_INPUT 64
enum { GYMRL_ACT_NONE = 0, GYMRL_ACT_TANH = 1, GYMRL_ACT_RELU = 2, GYMRL_ACT_CLAMP = 3, GYMRL_ACT_DUELING = 4, GYMRL_ACT_SILU = 5 /* 3-5: gymrl_lin_* only; 4, 5: forward only */ };
typedef struct {
  const float* W;
  const float* b;
  float* out;       /* dst == -1 only */
  int in_dim, out_dim, act, src, dst, out_stride;
} gymrl_mlp_stage;
typedef struct {
  int n_stages;
  gymrl_mlp_stage stage[GYMRL_MLP_MAX_STAGES];
} gymrl_mlp_desc;
size_t gymrl_mlp_packed_floats(int out_dim, int in_dim);
int gymrl_mlp_pack(const float* W, int out_dim, int in_dim, float* packed, void* stream);
int gymrl_mlp_forward(const float* x, int n_rows, int in_dim, const gymrl_mlp_desc* desc,
                      void* stream);

/* ======================================================== persistent rollout */
/*
 * P4: PPOTrainer.collect_rollout — ppo_lunarlander.py:198-231 — for LunarLander-v3 as one launch per
 * chunk of vector steps: policy forward (gymrl_mlp_forward's stages, logits [4] then value [1] as the two
 * dst == -1 stages; their `out` pointers are ignored), categorical draw, GAE chunk-map composition, env
 * step with reset-on-done, slab writes.  A workgroup owns 16 envs for all `nsteps` steps and never waits
 * for another workgroup, so a step costs the mean per-wavefront solver time instead of the slowest
 * wavefront's.  Bit-identical to the step-by-step sequence gymrl_mlp_forward -> gymrl_categorical_sample
 * (same seed / counter0 + t / env ids, or the same explicit noise) -> gymrl_env_step.
 *   obs f32[T+1][N][8] (row t0 must hold the current observations), act i32[T][N], logp/val/rew f32[T][N],
 *   done u8[T][N], ep_ret f32[T][N] or NULL (written where done), next_value f32[N] (written by the
 *   launch that reaches t0 + nsteps == T: V(obs[T]), the bootstrap of :225-229), noise_exp f32[T][N][4] or
 *   NULL, gae_running f64[2][N] + gae_workspace (gymrl_gae_workspace_bytes(T, N)) or NULL for no online
 *   composition (then gymrl_gae variant 1), ep_stats f64[3] or NULL, wg_ticks: optional load-balance probe.
 */
typedef struct {
  void* env_state;            /* gymrl_env_state_bytes(GYMRL_ENV_LUNARLANDER, n_envs)          */
  int n_envs;
  uint64_t seed;
  int64_t env_id0;
  uint64_t counter0;          /* Philox counter of step 0 of this rollout (step t uses counter0 + t) */
  float* obs;
  int32_t* act;
  float* logp;
  float* val;
  float* rew;
  uint8_t* done;
  float* ep_ret;
  float* next_value;
  const float* noise_exp;
  double* gae_running;
  void* gae_workspace;
  double gamma, lam;
  double* ep_stats;
  unsigned long long* wg_ticks; /* NULL, or u64[2][ceil(N/16)]: start / end of every workgroup in 100 MHz ticks */
  int T, t0, nsteps;
  /* gymrl_rollout_lunar_mhc only (gymrl_rollout_lunar ignores them): */
  float* ent;                 /* f32[T][N] or NULL: entropy of the behaviour policy at every step             */
  double lam2;                /* decoupled-lambda GAE (G3): the critic's lambda, with gae_running2             */
  double* gae_running2;       /* f64[2][N] or NULL; with it gae_workspace is gymrl_gae_decoupled_workspace_bytes(T, N) */
  /* both kernels: */
  int refill;                 /* != 0: while wave 0 steps the envs, wave 1 of the workgroup keeps every env's NEXT episode
                               * prepared in the state buffer's spare world (reset() ends with a full physics step; built inline
                               * it stalls the 15 other envs of the wave on every step an episode ends).  A spare is a pure
                               * function of (seed, env id, episode): the slab is bit-identical either way.             */
  /* gymrl_rollout_lunar only (ABI 3): */
  int gae_carry;              /* != 0 (with gae_running): the launch that reaches T also runs the blocked scan's carry pass for its
                               * own envs into the workspace (gymrl_gae_blk_carry's arithmetic, operation for operation), so
                               * gymrl_gae may be called with variant 3 — the apply launch alone; the same bits as variant 2 */
} gymrl_rollout_lunar_args;
int gymrl_rollout_lunar(const gymrl_rollout_lunar_args* args, const gymrl_mlp_desc* policy, void* stream);
/* The same rollout for PPO on CartPole-v1 (PPOTrainer with env_name "CartPole-v1": the reference's collect_rollout is
 * env-agnostic, ppo_lunarlander.py:198-231): obs f32[T+1][N][4], the policy's two dst == -1 stages are logits [2] then
 * value [1], env_state is gymrl_env_state_bytes(GYMRL_ENV_CARTPOLE, n_envs); `ent`, `lam2`, `gae_running2`, `refill` and
 * `wg_ticks` are ignored.  A step by step CartPole vector step is three launches at the launch floor; here a workgroup
 * keeps its 16 envs for the whole chunk.  Bit-identical to gymrl_mlp_forward -> gymrl_categorical_sample -> gymrl_env_step. */
int gymrl_rollout_cartpole(const gymrl_rollout_lunar_args* args, const gymrl_mlp_desc* policy, void* stream);
/* The same persistent rollout for PPO-full (ppo_full_lunarlander.py collect_experience :440-505): the policy is the mHC network
 * (gymrl_mhc_policy, declared with gymrl_mhc_policy_forward below; obs_dim 8, n_act 4), `ent` receives the behaviour policy's
 * entropy, and with gae_running2 / lam2 both decoupled-lambda chunk maps are composed (the actor's with `lam`, then the critic's).
 * Bit-identical to the step-by-step sequence gymrl_mhc_policy_forward -> gymrl_categorical_sample(online) -> gymrl_env_step. */
struct gymrl_mhc_policy_s;
int gymrl_rollout_lunar_mhc(const gymrl_rollout_lunar_args* args, const struct gymrl_mhc_policy_s* policy, void* stream);

/* ================================================ low-latency Linear layers == */
/*
 * D2 / A2 / T1 update path and the vectorised acting forward of the off-policy trainers: every
 * nn.Linear (+ ReLU / Tanh / clamp) of QNetwork dqn_cartpole.py:56-65, DuelingNoisyNetwork
 * rainbow_dqn_cartpole.py:97-113, Actor / Critic sac_pendulum.py:49-125, td3_pendulum.py:49-83,
 * ddpg_pendulum.py:37-48, sac_cartpole.py:47-70 — forward as called in update() / select_action(), and the
 * backward loss.backward() runs through it.  The reference issues a GEMM, a bias add, an activation, and in
 * backward an activation-gradient, two GEMMs and a bias reduction per layer; here a layer is ONE launch per
 * direction (csrc/lin.hip: one wavefront per 16-row output tile, v_mfma_f32_16x16x4_f32, operands from L2),
 * for batches up to a few thousand rows.  Up to GYMRL_LIN_MAX_ITEMS layers of one shape share a launch
 * (twin critics, mean / log_std heads, online + target network); the activation is per item.
 *
 *   fwd         y  = act(cat(x, x2) w^T + b)       x [B, K1] (row stride ldx), x2 [B, K - K1] (ldx2; K1 == K: unused),
 *                                                  w [N, K] row-major = nn.Linear.weight, b [N] or NULL, y [B, N] (ldy)
 *   bwd_input   (dx | dx2) (+)= (dy * act'(y)) w   dy, y [B, N] (ldy); dx [B, K1] (lddx), dx2 [B, K - K1] (lddx2);
 *                                                  a NULL dx / dx2 skips that part (at least one is required);
 *                                                  sum_items != 0: ONE result, the sum over the items (layers fed by the
 *                                                  same input), written to item 0's dx / dx2
 *   bwd_weight  dw (+)= (dy * act'(y))^T cat(x, x2),  db (+)= column sums of dy * act'(y)   (db NULL: skipped)
 *
 * act' is taken from the saved OUTPUT y: ReLU y > 0; Tanh 1 - y^2 (tanh = 1 - 2/(exp(2z)+1) on the exp2/rcp
 * units, as in gymrl_linear_fwd); clamp(lo, hi) lo < y < hi (torch passes the gradient AT the bounds as well — a
 * measure-zero difference).  accumulate != 0 adds to the destination (torch's .grad +=).  Sums are f32 fma chains
 * in the order given at the top of csrc/lin.hip; bwd_weight cuts B > 512 into <= 16 row slices whose partial
 * results are added in slice order by a second launch (workspace: gymrl_lin_workspace_bytes, else NULL).
 * Compared with torch float64 at 1e-5 relative (tests/test_lin_gpu.py), not bit for bit.
 */
#define GYMRL_LIN_MAX_ITEMS 4
typedef struct {
  const float* x;
  const float* x2;
  const float* w;
  const float* b;
  float* y;           /* fwd: output; bwd: the saved output (NULL allowed when act == GYMRL_ACT_NONE) */
  const float* dy;
  float* dx;
  float* dx2;
  float* dw;
  float* db;
  int act;            /* GYMRL_ACT_*; clamp bounds in lo / hi */
  float lo, hi;
  int32_t* argmax;    /* fwd with GYMRL_ACT_DUELING only: NULL, or i32[B] = first index of the row maximum of q (the greedy
                         action, torch.argmax's tie rule) */
} gymrl_lin_item;
size_t gymrl_lin_workspace_bytes(int B, int N, int K, int n_items);
int gymrl_lin_fwd(const gymrl_lin_item* items, int n_items, int B, int K, int K1, int N, int ldx, int ldx2, int ldy,
                  void* stream);
int gymrl_lin_bwd_input(const gymrl_lin_item* items, int n_items, int B, int N, int K, int K1, int ldy, int lddx,
                        int lddx2, int accumulate, int sum_items, void* stream);
int gymrl_lin_bwd_weight(const gymrl_lin_item* items, int n_items, int B, int N, int K, int K1, int ldy, int ldx,
                         int ldx2, int accumulate, void* workspace, void* stream);

/*
 * gymrl_lin_fwd with act == GYMRL_ACT_DUELING (N = A + 1 <= 16 stacked rows: A advantage rows, then the value row)
 * writes q [B, A] = value + advantage - mean(advantage) — DuelingNoisyNetwork.forward rainbow_dqn_cartpole.py:108-113 —
 * instead of the N raw columns; gymrl_dueling_bwd maps dq [B, A] back to the stacked dS [B, A + 1].
 *
 * NoisyLinear (rainbow_dqn_cartpole.py:60-95): gymrl_noisy_combine stacks the effective parameters of up to
 * GYMRL_NOISY_MAX_LAYERS layers that share K — W[row] = w_mu + w_sigma * w_eps, b likewise (training == 0: mu only) —
 * into W_out [sum n_out, K] / b_out [sum n_out] and copies the noise through to w_eps_copy / b_eps_copy (the
 * module's weight_epsilon / bias_epsilon buffers) when given; gymrl_noisy_split sends the stacked gradient back:
 * d mu (+)= dW, d sigma (+)= dW * eps (zeros when training == 0; NULL d*_sigma: skipped).
 */
#define GYMRL_NOISY_MAX_LAYERS 8
typedef struct {
  const float* w_mu; const float* w_sigma; const float* w_eps;      /* [n_out, K] */
  const float* b_mu; const float* b_sigma; const float* b_eps;      /* [n_out] */
  float* w_eps_copy; float* b_eps_copy;                             /* combine only, NULL: no copy */
  float* dw_mu; float* dw_sigma; float* db_mu; float* db_sigma;     /* split only */
  uint64_t seed, counter;                                           /* combine with draw != 0 */
  const uint64_t* counter_dev;                                      /* NULL, or the counter in device memory (graphs) */
  int draw;            /* combine: draw this layer's noise in the launch itself — bit for bit gymrl_noisy_noise(seed,
                          counter) — instead of reading w_eps / b_eps; it is written to w_eps_copy / b_eps_copy when given */
  int eval;            /* combine: this layer contributes its mu only even when training != 0 (a target network's layer
                          stacked beside training-mode ones) */
  int n_out;
} gymrl_noisy_layer;
int gymrl_noisy_combine(const gymrl_noisy_layer* layers, int n_layers, int K, int training, float* W_out, float* b_out,
                        void* stream);
int gymrl_noisy_split(const gymrl_noisy_layer* layers, int n_layers, int K, int training, const float* dW, const float* db,
                      int accumulate, void* stream);
/* MFMA-operand images of a square Linear weight W [H][H] (H % 16 == 0), csrc/lin_device.hpp: the forward operand (lane (r, q) of
 * (tile t, step c) holds W[16t + r][16c + 4q .. + 3]) and the input-gradient operand (W[16c + 4q .. + 3][16t + r]), one contiguous
 * 1 KiB block per wave-wide load — what a slab kernel streams 2-3x faster than nn.Linear's rows.  Either output may be NULL. */
typedef struct { const float* W; int H; float* img_fwd; float* img_bwd; } gymrl_weight_image;
#define GYMRL_NOISY_MAX_IMAGES 4
/* gymrl_noisy_combine and, on extra workgroups of the SAME launch, n_images (<= GYMRL_NOISY_MAX_IMAGES) weight images rebuilt from
 * the parameters as they are: Rainbow's vector step has this launch between the optimiser step and the acting forward anyway. */
int gymrl_noisy_combine_images(const gymrl_noisy_layer* layers, int n_layers, int K, int training, float* W_out, float* b_out,
                               const gymrl_weight_image* images, int n_images, void* stream);
int gymrl_dueling_bwd(const float* dq, int B, int A, float* dS_out, void* stream);

/* ================================================ mHC backbone ============== */
/*
 * F1: ManifoldHyperConnectionFuse.gates + MHCBlock._sub + RMSNorm of PPO-full's network — ppo_full_lunarlander.py:106-194
 * (gates :125-147, sub-block :160-165), RMSNorm :96-104, MHCBackbone.forward :178-183 — as called from get_action :395-407 /
 * get_value once per env step and from the minibatch passes of update_model :537-660.  The reference issues ~95 torch launches per
 * hyper-connection; here a sub-block's forward is gymrl_mhc_gates + gymrl_lin_fwd + gymrl_mhc_combine.
 *   gymrl_mhc_gates: h [B, n, D] (n = 2 or 4 branches) -> pre [B, n] = sigmoid(r H[:n] a0 + beta), post [B, n] =
 *     2 sigmoid(r H[n:2n] a1 + beta), mix [B, n, n] = u A v with A = exp(r H[2n:] a2 + beta) and u, v from sk_it
 *     Sinkhorn-Knopp sweeps, where H = (norm_w * flat) w, r = 1 / (|flat| / sqrt(nD) + 1e-6); read [B, D] = sum_i pre_i h_i.
 *     norm_w [nD] = fuse.norm.weight, w [nD, n*n + 2n], alpha [3], beta [n*n + 2n].  stats_out (nullable; n = 2 and
 *     n*D = 256 or 512 only): f32[B, 9] = the row's eight read-out sums H and |flat|^2, the input of gymrl_mhc_gates_bwd.
 *   gymrl_mhc_combine: h_out[b, i, :] = post[b, i] o[b, :] + sum_j mix[b, i, j] h[b, j, :], o = out (act = GYMRL_ACT_NONE) or
 *     SiLU(out) (GYMRL_ACT_SILU: `out` is the Linear's raw output, the training pass keeps it for the backward).
 *   gymrl_rmsnorm: y [B, D] = s rsqrt(mean(s^2) + eps) w, s = the sum of the row's n_sum consecutive [D] blocks
 *     (n_sum = n: final_norm(h.sum(1)); 1: the MLPs' RMSNorm), through SiLU first when act = GYMRL_ACT_SILU
 *     (the MLPs' Linear -> SiLU -> RMSNorm :371-402).
 * Floating point, compared with the torch modules at 1e-5 (tests/test_mhc_fused_gpu.py).
 */
int gymrl_mhc_gates(const float* h, const float* norm_w, const float* w, const float* alpha, const float* beta, int B, int n,
                    int D, int sk_it, float* pre_out, float* post_out, float* mix_out, float* read_out, float* stats_out,
                    void* stream);
int gymrl_mhc_combine(const float* post, const float* mix, const float* out, const float* h, int B, int n, int D, int act,
                      float* h_out, void* stream);
int gymrl_rmsnorm(const float* x, const float* w, int B, int D, int n_sum, float eps, int act, float* y, void* stream);
/* Backward of gymrl_rmsnorm (n_sum = 1, D <= 512): g = dL/dy -> d_x [B, D] (dL/dx, through SiLU' when act = GYMRL_ACT_SILU)
 * and d_w [D] (overwritten; per-workgroup partial sums in `workspace`, gymrl_rmsnorm_bwd_workspace_bytes, added in a fixed
 * order — no atomics). */
size_t gymrl_rmsnorm_bwd_workspace_bytes(int D);
int gymrl_rmsnorm_bwd(const float* g, const float* x, const float* w, int B, int D, float eps, int act, float* d_x, float* d_w,
                      void* workspace, void* stream);
/* The same for the norm of a branch sum (final_norm(h.sum(dim=1)), MHCBackbone.forward :243): x [B, n_sum, D], d_x [B, D] is
 * the gradient of EVERY block (the sum hands each the same one — the caller broadcasts it, nothing is materialised). */
int gymrl_rmsnorm_sum_bwd(const float* g, const float* x, const float* w, int B, int D, int n_sum, float eps, int act, float* d_x,
                          float* d_w, void* workspace, void* stream);
/* A head's tail in one launch each way: out [B, n_out] = RMSNorm(SiLU(x)) W2^T + b2 for x [B, D], D <= 256, n_out <= 8
 * (MLP([128, 256, n_out]) :371-402: the actor's and the critic's Linear -> SiLU -> RMSNorm -> Linear; b2 may be NULL), and its
 * backward from d_out [B, n_out]: d_x [B, D], d_norm_w [D], d_W2 [n_out, D], d_b2 [n_out] (overwritten; per-workgroup partial
 * sums in `workspace`, added in a fixed order).  The normalised activations and their gradient never touch HBM.
 * The kernel (one row or four rows per wave, i.e. the summation order) is chosen by (D, n_out) ALONE: at D = 256 x, norm_w, W2
 * (and d_x, n_out = 1) must be 16-byte aligned — -22 otherwise; a result's bits never depend on where a buffer starts. */
int gymrl_norm_proj_fwd(const float* x, const float* norm_w, const float* W2, const float* b2, int B, int D, int n_out, float eps,
                        float* out, void* stream);
size_t gymrl_norm_proj_bwd_workspace_bytes(int D, int n_out);
int gymrl_norm_proj_bwd(const float* d_out, const float* x, const float* norm_w, const float* W2, int B, int D, int n_out, float eps,
                        float* d_x, float* d_norm_w, float* d_W2, float* d_b2, void* workspace, void* stream);
/* The Sinkhorn-Knopp sweeps alone (:141-146; constants of the backward pass in the reference): A f32[B, n, n] > 0 ->
 * u [B, n], v [B, n] after sk_it sweeps u = 1/(A v + 1e-8), v = 1/(A^T u + 1e-8) from u = v = 1 — for gate shapes
 * gymrl_mhc_gates_bwd does not cover, whose other gate operations stay with autograd. */
int gymrl_sinkhorn(const float* A, int B, int n, int sk_it, float* u_out, float* v_out, void* stream);
/* Training pass of MHCBlock._sub (:160-165): the branch products' backward (autograd wraps them:
 * gymrl_amd/ppo_full_lunarlander.py _MhcSub; _MhcRead / _MhcCombine for the shapes it does not cover).
 *   read_fwd:     read [B, D] = sum_i pre[b, i] h[b, i, :]
 *   read_bwd:     d_pre [B, n] = sum_d g[b, d] h[b, i, d];  d_h [B, n, D] (+)= pre[b, i] g[b, d]   (d_h NULL: d_pre only)
 *   combine_bwd (forward = gymrl_mhc_combine with the same act), g = dL/dh' [B, n, D]:
 *                 d_post [B, n], d_mix [B, n, n], d_out [B, D] (act = GYMRL_ACT_SILU: `out` is the raw Linear output z and
 *                 d_out is dL/dz), d_h [B, n, D] = mix^T g (overwritten; NULL: left to gymrl_mhc_gates_bwd's g_out term) */
/* Backward of the gates (n = 2, n*D = 256 or 512; forward = gymrl_mhc_gates with stats_out): given dL/d pre, post, mix — u, v of
 * the Sinkhorn sweeps are constants, as in the reference — writes d_h [B, n, D] (overwritten) and the parameter gradients
 * d_norm_w [nD], d_w [nD, n*n + 2n], d_alpha [3], d_beta [n*n + 2n] (overwritten; sums over rows in a fixed order:
 * per-workgroup partial vectors in `workspace`, gymrl_mhc_gates_bwd_workspace_bytes, added ascending — no atomics).
 * d_read (nullable, [B, D]) and g_out (nullable, [B, n, D]) fold the sub-block's other two paths into d_h in the same pass:
 * d_h[b, j] += pre[b, j] d_read[b] (the read's backward) + sum_i mix[b, i, j] g_out[b, i] (the combine's), so that the three
 * consumers of h produce ONE gradient tensor and autograd adds nothing. */
size_t gymrl_mhc_gates_bwd_workspace_bytes(int n, int D);
int gymrl_mhc_gates_bwd(const float* h, const float* norm_w, const float* w, const float* alpha, const float* pre, const float* post,
                        const float* mix, const float* stats, const float* d_pre, const float* d_post, const float* d_mix,
                        const float* d_read, const float* g_out, int B, int n, int D, float* d_h, float* d_norm_w, float* d_w,
                        float* d_alpha, float* d_beta, void* workspace, void* stream);
int gymrl_mhc_read_fwd(const float* pre, const float* h, int B, int n, int D, float* read_out, void* stream);
int gymrl_mhc_read_bwd(const float* g, const float* pre, const float* h, int B, int n, int D, float* d_pre, float* d_h,
                       int accumulate, void* stream);
int gymrl_mhc_combine_bwd(const float* g, const float* post, const float* mix, const float* out, const float* h, int B, int n, int D,
                          int act, float* d_post, float* d_mix, float* d_out, float* d_h, void* stream);

/* Training pass: the forward of one hyper-connection sub-block (MHCBlock._sub :160-165 with the gates :125-147) in ONE launch for
 * n = 2 branches of D = 128: h [B, 2, 128] -> h_out = post (x) SiLU(z) + mix h with z = read lin_w^T + lin_b, read = sum_i pre_i h_i,
 * and everything the backward takes: pre_out [B, 2], post_out [B, 2], mix_out [B, 2, 2], stats_out [B, 9] (as gymrl_mhc_gates),
 * read_out [B, 128], z_out [B, 128].  h_broadcast != 0: h is [B, 128], the same row for both branches (the first sub-block's
 * input is the input projection repeated, MHCBackbone.forward :239: no [B, 2, 128] copy is made of it).  The same values as gymrl_mhc_gates + gymrl_lin_fwd + gymrl_mhc_combine(GYMRL_ACT_SILU) up to
 * the order of the Linear's sums; 16-row tiles per wave, the weights staged in LDS once per workgroup. */
int gymrl_mhc_sub_forward(const float* h, int h_broadcast, const float* norm_w, const float* w, const float* alpha, const float* beta,
                          const float* lin_w, const float* lin_b, int B, int n, int D, int sk_it, float* pre_out, float* post_out,
                          float* mix_out, float* stats_out, float* read_out, float* z_out, float* h_out, void* stream);
/* Training pass: the backward of the same sub-block in ONE launch (+ the fixed-order reduction of the parameter partials).
 * Inputs: the upstream gradient g [B, 2, 128] (g_broadcast != 0: [B, 128], the same row for both branches — the final norm's
 * d x), the saved h [B, 2, 128] (h_broadcast != 0: [B, 128], the repeated input of the first sub-block), z, pre, post, mix, stats.
 * Outputs: d_z [B, 128] = dL/dz (what gymrl_lin_bwd_weight takes with `read` for the Linear's d W, d b), d_h [B, 2, 128] — the
 * gates', the read's and the combine's paths into h added (sum_branches != 0: [B, 128], the two branches' gradients added:
 * the gradient of a repeated row), d_norm_w [256], d_w [256, 8], d_alpha [3], d_beta [8].  d_z and d_h must have room for
 * ceil(B / 16) * 16 rows: the kernel writes whole 16-row tiles (rows past B hold unspecified values).  workspace:
 * gymrl_mhc_gates_bwd_workspace_bytes(2, 128).  The values of gymrl_mhc_combine_bwd + gymrl_linear_bwd_input +
 * gymrl_mhc_read_bwd + gymrl_mhc_gates_bwd up to the order of the sums (float64 autograd at 3e-5: tests/test_mhc_fused_gpu.py);
 * g and h cross HBM once instead of three times. */
int gymrl_mhc_sub_backward(const float* g, int g_broadcast, const float* h, int h_broadcast, const float* z, const float* pre,
                           const float* post, const float* mix, const float* stats, const float* norm_w, const float* w,
                           const float* alpha, const float* lin_w, int B, int n, int D, float* d_z, float* d_h, int sum_branches,
                           float* d_norm_w, float* d_w, float* d_alpha, float* d_beta, void* workspace, void* stream);
/* The whole rollout forward of PPO-full's network (ActorCritic.forward :377-407 as called by get_action / get_value) in ONE
 * launch, for the reference's default shape: n = 2 branches of D = 128 (mhc_rate, mhc_dim), 256-wide heads
 * (MLP([128, 256, n_out]) :371-402), obs_dim <= 16, n_act <= 8, n_sub = 2 * mhc_layers <= 8 sub-blocks.  Rows are
 * independent, so 16 of them travel through every layer inside one workgroup (branch stack in registers, the Linears on
 * v_mfma_f32_16x16x4_f32).  Pointers are the modules' own parameters (nn.Linear layout [out, in]); logits_out [B, n_act],
 * value_out [B].  Same values as the per-layer entry points above to 1e-5 (tests/test_mhc_fused_gpu.py). */
typedef struct {
  const float* norm_w;   /* fuse.norm.weight [256] */
  const float* w;        /* fuse.w [256, 8] */
  const float* alpha;    /* [3] */
  const float* beta;     /* [8] */
  const float* lin_w;    /* the sub-block's Linear [128, 128] */
  const float* lin_b;    /* [128] */
} gymrl_mhc_sub;
typedef struct {
  const float* w1;       /* mlp.0.weight [256, 128] */
  const float* b1;       /* [256] */
  const float* norm_w;   /* mlp.2.weight [256] */
  float norm_eps;
  const float* w2;       /* mlp.3.weight [n_out, 256] (n_out = n_act for head 0, 1 for head 1) */
  const float* b2;       /* [n_out] */
} gymrl_mhc_head;
typedef struct gymrl_mhc_policy_s {
  int obs_dim, n_sub, n_act, sk_it;
  const float* in_w;     /* input_proj.weight [128, obs_dim] */
  const float* in_b;     /* [128] */
  gymrl_mhc_sub sub[8];
  const float* final_norm_w;   /* [128] */
  float final_norm_eps;
  gymrl_mhc_head head[2];      /* 0 = actor, 1 = critic */
  const float* image;          /* NULL: every weight is read in place.  Else gymrl_mhc_policy_pack's output (16-byte aligned,
                                * gymrl_mhc_policy_image_floats(n_sub) floats): the sub-blocks' Linears and gate weights and the heads'
                                * first Linears in the order the kernel's lanes read them — one contiguous KiB per wave-wide load
                                * instead of a 64-byte piece of sixteen rows; repack after the parameters change.  Same values. */
} gymrl_mhc_policy;
size_t gymrl_mhc_policy_image_floats(int n_sub);
int gymrl_mhc_policy_pack(const gymrl_mhc_policy* p, float* image, void* stream);      /* p->image is ignored */
int gymrl_mhc_policy_forward(const gymrl_mhc_policy* p, const float* obs, int B, float* logits_out, float* value_out, void* stream);

/* ===================================================== MLP update path ===== */
/*
 * The HBM-bound passes of one ActorCritic minibatch update around the library GEMMs —
 * ppo_lunarlander.py:274-307: evaluate_actions' forward :110-117 (shared/actor/critic
 * Sequentials :67-84) and loss.backward() :303.  Activations are row-major f32 [B, C],
 * C a power of two in 16..256.  workspace: gymrl_mlp_train_workspace_bytes(C, D, A) bytes.
 * Column reductions are deterministic (fixed-order f64 finalize over per-workgroup partials).
 *
 *   linear_tanh_smallk : out = tanh(x W^T + b), x [B, D], W [C, D], D in {2,3,4,8}
 *                        (shared.0: Linear(obs, hidden) + Tanh)
 *   tanh_inplace       : z <- tanh(z + bias[col]) on n floats = rows of C columns (bias NULL: none) after a
 *                        library GEMM; the bias rides on this pass, the GEMM runs without an epilogue
 *   tanh_bwd_colsum    : dH <- dH * (1 - H^2) in place; colsum_out[c] = sum_r dH[r][c]
 *                        (Tanh backward + the bias gradient of the Linear below it)
 *   linear_smallk_bwd  : dZ = dH * (1 - H^2) (never stored); dW [C, D] = dZ^T x; db [C] = colsum dZ.
 *                        W, b non-NULL (the layer's own weight [C, D] / bias): H is not read (may be NULL) —
 *                        H = tanh(x W^T + b) is recomputed exactly as linear_tanh_smallk computed it
 *   heads_fwd_tanh     : Zac [B, 2C] = pre-activations of actor.0 | critic.0 (one N = 2C GEMM; their biases
 *                        bac [2C] are added here, NULL if the GEMM already did) -> Hac = tanh(Zac + bac) in place AND logits [B, A] = Ha Wa2^T + ba2,
 *                        value [B] = Hc Wc2^T + bc2 from the tanh values still in registers
 *                        (actor.0/critic.0's Tanh + actor.2 + critic.2 forward in one pass).
 *                        store_h == 0: Zac is NOT rewritten (it keeps the pre-activations; 2 KB/row less HBM
 *                        traffic) — then call heads_bwd with pre_activation = 1, which recomputes the same tanh
 *   heads_bwd          : Hac [B, 2C] = [Ha | Hc], the Tanh outputs of actor.0 / critic.0;
 *                        dlogits [B, A] (A in {2,4}), dv [B] from gymrl_ppo_loss_fwd_bwd;
 *                        Wa2 [A, C] = actor.2.weight, Wc2 [1, C] = critic.2.weight.  Writes
 *                        dZac [B, 2C] = [(dlogits Wa2)(1 - Ha^2) | (dv Wc2)(1 - Hc^2)],
 *                        dbac [2C] = colsum dZac, dWa2 = dlogits^T Ha, dba2 = colsum dlogits,
 *                        dWc2 = dv^T Hc, dbc2 = sum dv — one pass over Hac instead of four
 *                        skinny GEMMs, two tanh' passes and four reductions.
 *                        pre_activation != 0: `Hac` holds the pre-activations instead and the kernel applies
 *                        tanh(. + bac[col]) (bac NULL: no bias) first — bit for bit the forward's values
 * tanh here (and in gymrl_linear_fwd's epilogue) is 1 - 2/(exp(2x)+1) on the hardware exp2/rcp units:
 * device-deterministic, absolute error < 2e-7, exact +-1 limits; compared with torch at 1e-5, not bit for bit.
 * linear_smallk_bwd with H == NULL and W == NULL takes dH as dZ itself (the tanh' factor was applied upstream,
 * gymrl_linear_bwd_input).
 */
size_t gymrl_mlp_train_workspace_bytes(int C, int D, int A);
int gymrl_linear_tanh_smallk(const float* x, const float* W, const float* b, int64_t B, int D, int C,
                             float* out, void* stream);
/* The same launch without the tanh: out [B, C] = x [B, D] W[C, D]^T + b (D in {2, 3, 4, 8}, C a power of two >= 4), one fmaf chain
 * per output in ascending d — PPO-full's input projection (ppo_full_lunarlander.py:236: Linear(obs, 128)) at 524 288-row
 * micro-batches, where the 16 x 16-tile layer kernel writes its 268 MB in 64-byte pieces (152 us; this one: 16-byte stores). */
int gymrl_linear_smallk(const float* x, const float* W, const float* b, int64_t B, int D, int C, float* out, void* stream);
int gymrl_tanh_inplace(float* z, int64_t n, const float* bias, int C, void* stream);
int gymrl_tanh_bwd_colsum(float* dH, const float* H, int64_t B, int C, float* colsum_out,
                          void* workspace, void* stream);
int gymrl_linear_smallk_bwd(const float* dH, const float* H, const float* x, int64_t B, int D, int C,
                            float* dW, float* db, const float* W, const float* b, void* workspace,
                            void* stream);
int gymrl_heads_fwd_tanh(float* Zac, int64_t B, int C, int A, const float* bac, const float* Wa2,
                         const float* ba2, const float* Wc2, const float* bc2, float* logits, float* value,
                         int store_h, void* stream);
int gymrl_heads_bwd(const float* Hac, const float* dlogits, const float* dv, int64_t B, int C, int A,
                    const float* Wa2, const float* Wc2, float* dZac, float* dbac, float* dWa2,
                    float* dba2, float* dWc2, float* dbc2, int pre_activation, const float* bac,
                    void* workspace, void* stream);

/*
 * heads_loss_fwd_bwd : gymrl_heads_fwd_tanh + gymrl_ppo_loss_fwd_bwd + gymrl_heads_bwd as ONE pass over
 *   Zac [B, 512] (C == 256): per row tanh(Zac + bac), logits / value from the two heads, the clipped-surrogate loss of
 *   ppo_lunarlander.py:278-300 (advantage normalised on load from adv_moments as in gymrl_ppo_loss_fwd_bwd; act /
 *   logp_old / adv / ret are the minibatch's rows in order, i.e. already gathered), and the heads' backward: dZac
 *   overwrites Zac in place; dbac [512], dWa2 [A, 256], dba2 [A], dWc2 [256], dbc2 [1] as gymrl_heads_bwd.
 *   metric_parts: f64[gymrl_heads_loss_blocks(B, C)][5] block partials of (policy loss, value loss, entropy,
 *   clip fraction, approx KL) sums — reduce with gymrl_reduce_rows.  Same arithmetic per row as the three
 *   separate passes (one shared device function each), so gradients and metrics are bit-identical to them.
 */
/* (gradient outputs: all five non-NULL, or all five NULL = "block partials only": they stay in `workspace` for
 *  gymrl_update_finalize below; gymrl_linear_smallk_bwd takes dW = db = NULL and gymrl_linear_bwd_weight dW = NULL the same way) */
int gymrl_heads_loss_blocks(int64_t B, int C);
int gymrl_heads_loss_fwd_bwd(float* Zac, int64_t B, int C, int A, const float* bac, const float* Wa2,
                             const float* ba2, const float* Wc2, const float* bc2, const int32_t* act,
                             const float* logp_old, const float* adv, const float* ret,
                             const double* adv_moments, const gymrl_ppo_cfg* cfg, float* dbac, float* dWa2,
                             float* dba2, float* dWc2, float* dbc2, double* metric_parts, void* workspace,
                             void* stream);

/* ===================================================== update-path GEMMs === */
/*
 * The 256-wide contractions of one ActorCritic minibatch update — ppo_lunarlander.py:274-307:
 * shared.2 / actor.0 / critic.0 of evaluate_actions' forward (:67-84, :110-117) and their
 * input / weight gradients under loss.backward() (:303) — as hand-written exact-f32 MFMA kernels
 * (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit for bit an fmaf chain; the reference computes
 * in f32 and gfx950 has no TF32) with the layer's elementwise work in the epilogue.  Row-major f32,
 * K = 256 (hidden_dim), B rows.  workspace: gymrl_gemm_workspace_bytes() bytes, 16-B aligned.
 *
 *   linear_fwd         Y [B, N] = act(X [B, K] W[N, K]^T + b[N]);  N in {256, 512}; act 0 none, 1 tanh
 *                      (b NULL: no bias).  Linear + Tanh of shared.2 (N = 256) and of actor.0 | critic.0
 *                      as one N = 512 layer over their adjacent weights.
 *   linear_bwd_input   dX [B, K] = (dY [B, N] W[N, K]) * (1 - H^2), H [B, K] = the tanh output of the layer
 *                      BELOW (H NULL: no factor).  N in {256, 512}.  (The bias gradient of the layer below —
 *                      the column sums of dX — comes out of that layer's linear_bwd_weight as `db`.)
 *   linear_bwd_weight  dW [N, K] = dY [B, N]^T X [B, K];  N in {256, 512}.
 *
 * Accumulation order (part of the contract — oracle/gymrl_oracle.c restates it and tests compare bit for
 * bit).  fwd / bwd_input: one fmaf chain per output from +0 over the reduction index r in chunks of 8
 * ascending, inside a chunk in the order 0, 4, 1, 5, 2, 6, 3, 7; bias / factor applied after the chain in
 * f32 ((acc + b), acc * (1 - h*h)).  bwd_weight: the rows are cut into `slices` slices of
 * `rows_per_slice` rows (gymrl_linear_bwd_weight_geometry); inside a slice one fmaf chain from +0 over
 * the rows ascending; dW = ((g0 + g1) + g2) + g3 in f64, g_j = the f64 sum of the slice results s = j
 * (mod 4) ascending, rounded once to f32.  act = tanh uses the same hardware-exp2 form as the passes above.  bwd_weight's db [N]
 * (NULL: skipped) = column sums of dY — the bias gradient of the same layer: per slice the even- and the odd-offset
 * rows are summed sequentially in f32 and added, slices combine like the weight tiles (bit for bit in the oracle).
 * gymrl_gemm_config(key, value) exists ONLY in the probe build (make -C gymrl_amd/csrc prof -> libgymrl_hip_prof.so,
 * -DGYMRL_PROF_BUILD): it selects timing-only ablation variants of the kernels for tools/abl_gemm.py (4: bwd_weight,
 * 5: fwd; 0 = the product kernel).  The product library exports no such switch and keeps no mutable global state.
 */
size_t gymrl_gemm_workspace_bytes(void);
#ifdef GYMRL_PROF_BUILD
int gymrl_gemm_config(int key, int value);
#endif
int gymrl_linear_fwd(const float* X, const float* W, const float* b, int64_t B, int K, int N, int act,
                     float* Y, void* stream);
int gymrl_linear_bwd_input(const float* dY, const float* W, const float* H, int64_t B, int N, int K,
                           float* dX, void* stream);
/* dX [B, K] = G + dY W for the (N, K) = (256, 128) layer: the input gradient of a SECOND Linear on the same input added to the
 * first one's in the GEMM's epilogue (PPO-full's actor and critic heads both read the backbone's output: the separate add was a
 * 0.8 GB pass per 524 288-row micro-batch).  Same sums as gymrl_linear_bwd_input followed by G + . ; G may not alias dX. */
int gymrl_linear_bwd_input_add(const float* dY, const float* W, const float* G, int64_t B, int N, int K, float* dX, void* stream);
int gymrl_linear_bwd_weight_geometry(int64_t B, int N, int* slices, int64_t* rows_per_slice);
int gymrl_linear_bwd_weight(const float* dY, const float* X, int64_t B, int N, int K, float* dW,
                            float* db, void* workspace, void* stream);
/*
 * The second halves of the five batch reductions that close one ActorCritic minibatch backward (ppo_lunarlander.py:303) as ONE
 * launch: the slice partials of dWac = d[actor.0 | critic.0].weight (gymrl_linear_bwd_weight, N = 2C, called with dW = NULL)
 * and of dW2 / db2 = d shared.2 (N = C, dW = NULL, db != NULL), the block partials of the heads' gradients
 * (gymrl_heads_loss_fwd_bwd with its five outputs NULL) and of the first layer's (gymrl_linear_smallk_bwd, dW = db = NULL) —
 * each left in the workspace handed to that call (FOUR DIFFERENT workspaces: a deferred reduction's partials must survive
 * until this launch).  Every block runs the body of the kernel it replaces: same loads, same float64 sums in the same order,
 * outputs bit-identical to the five separate launches (tests/test_update_path_gpu.py).  C = 256, A in {2, 4}, D in {2, 3, 4, 8}.
 */
int gymrl_update_finalize(int64_t B, int C, int A, int D, const void* ws_dw_ac, float* dWac, const void* ws_dw_2, float* dW2,
                          float* db2, const void* ws_heads, float* dbac, float* dWa2, float* dba2, float* dWc2, float* dbc2,
                          const void* ws_smallk, float* dW1, float* db1, void* stream);

/* ========================================================= off-policy ===== */
/*
 * D2 / A3 / S2: device-resident replay ring, SoA rows [cap]:
 *   state f32[cap,D], action u32[cap,AW] (raw 4-byte words: i32 discrete action or
 *   f32 continuous components), reward f32[cap], next_state f32[cap,D], flag u8[cap]
 *   (done for DQN/SAC — dqn_cartpole.py:183, sac_pendulum.py:283; terminal for
 *   Rainbow — rainbow_dqn_cartpole.py:376-380).
 * Replaces ReplayBuffer.push/sample (dqn_cartpole.py:68-88, sac_pendulum.py:128-148)
 * and utils/buffer.py:105-135.  append writes rows (cursor + i) % cap, i < n.
 */
int gymrl_replay_append(float* state, uint32_t* action, float* reward, float* next_state,
                        uint8_t* flag, int64_t cap, int64_t cursor, int D, int AW, int n,
                        const float* src_state, const void* src_action, const float* src_reward,
                        const float* src_next_state, const uint8_t* src_flag, const int64_t* cursor_dev,
                        void* stream);
/* rows idx[b] -> contiguous batch (torch.tensor(np.array(...)) of dqn_cartpole.py:143-155) */
int gymrl_replay_gather(const float* state, const uint32_t* action, const float* reward,
                        const float* next_state, const uint8_t* flag, const int32_t* idx, int B,
                        int D, int AW, float* state_out, void* action_out, float* reward_out,
                        float* next_state_out, float* flag_out, void* stream);
/* random.sample(buffer, B) (dqn_cartpole.py:76) / np.random.choice(size, B, replace=False)
 * (utils/buffer.py:127): B DISTINCT uniform rows — idx[b] = the b-th element of a keyed permutation of
 * [0, size) (the Feistel/Philox bijection of gymrl_permutation, keyed by (seed, counter)); B <= size. */
int gymrl_uniform_indices(uint64_t seed, uint64_t counter, int64_t size, int B,
                          int32_t* idx_out, const void* dev, void* stream);

/*
 * S2: PrioritizedNStepBuffer.store_transition + _get_n_step_transition —
 * rainbow_dqn_cartpole.py:179-218 — for N envs at once.  Each env keeps its own
 * n-deep window (never cleared at episode end, :163) in caller-owned SoA arrays
 *   w_state f32[n,N,D], w_action i32[n,N], w_reward f32[n,N], w_next f32[n,N,D],
 *   w_terminal u8[n,N], w_done u8[n,N];
 * `pushes` = number of calls so far (slot = pushes % n).  When pushes + 1 >= n every env
 * emits one n-step row into the replay ring at (cursor + env) % cap:
 *   R = sum via R = r_i + gamma*(1-d_i)*R (float64, newest to oldest, :212-214),
 *   (next_state, terminal) from the newest entry unless some d_i, then from the
 *   EARLIEST done (:215-216).  Returns 1 if rows were emitted, 0 if still filling, <0 error.
 * terminal NULL: the flag is formed here as the reference's loop does (:376): done[e] && ep_len[e] != max_episode_steps
 * (ep_len i32[N] = the env's step index inside its episode, as gymrl_env_step reports it).
 */
int gymrl_nstep_push(float* w_state, int32_t* w_action, float* w_reward, float* w_next,
                     uint8_t* w_terminal, uint8_t* w_done, int n_steps, int64_t pushes,
                     int N, int D, double gamma,
                     const float* obs, const int32_t* action, const float* reward,
                     const float* next_obs, const uint8_t* terminal, const uint8_t* done,
                     float* r_state, uint32_t* r_action, float* r_reward, float* r_next,
                     uint8_t* r_flag, int64_t cap, int64_t cursor, const int64_t* dev, const int32_t* ep_len,
                     int max_episode_steps, void* stream);

/*
 * S1 / S1': SumTree — rainbow_dqn_cartpole.py:116-152 (variant A) and
 * ddqn_per_cartpole.py:67-104 (variant B).  tree f64[2*cap-1], leaf i at i+cap-1;
 * cap need not be a power of two (the array rule is followed literally).
 *   gymrl_per_update: the reference's `for idx, p in zip(...): tree.update(idx, p)`
 *     (:258-261 / :146-147): leaf := p, every ancestor += (p - old leaf), applied in
 *     batch order — duplicates resolve last-writer-wins and every node receives its
 *     additions in the reference's order, so the float64 array matches bit for bit.
 *     idx NULL -> the N-ROW VECTOR STORE: consecutive rows (idx_start + b) % cap (the store path :201-205), B <= cap.
 *       The reference stores one row per env step, so the order in which the N changes of a vector step reach an
 *       ancestor is defined HERE (and restated by the oracle's tree_store_chunk): change_b = p_b - old leaf, leaf := p_b;
 *       a complete binary tree over the batch index, seg[P + b] = change_b (+0.0 beyond B, P = the power of two >= B),
 *       seg[k] = seg[2k] + seg[2k + 1]; a node's elements (ascending b) form maximal runs of consecutive b; a run [a, e)
 *       is summed from the canonical blocks (l = a + P, r = e + P; while l < r: l odd -> sl += seg[l++]; r odd ->
 *       sr += seg[--r]; halve both; run = sl + sr); S = +0.0 + run_1 + run_2 ...; node := node + S — ONE addition per
 *       node, no chain longer than log2(P) + 4 float64 adds.  B > 8192: sub-stores of 8192 rows in order.  At B = 1 this
 *       is the reference's `tree[parent] += change` (:122-128) bit for bit; the strictly ordered per-node sums above
 *       remain for explicit index batches (update_priorities, whose order the reference does define).
 *     idx_is_tree != 0 -> idx are tree indices (variant B, :75-80).
 *     prio f64[B], or prio NULL and prio_scalar for all (store: 1.0 or priority_max).
 *     workspace >= gymrl_per_workspace_bytes(B).
 *   gymrl_per_max_leaf: priority_max (:151-152) = max over all leaves -> out f64[1].
 *   gymrl_per_priorities: p = min(|td| + eps, clip)^alpha in float64 (:259 eps 0.01,
 *     clip inf; ddqn_per_cartpole.py:142-145 eps 1e-4, clip 1.0).
 *   gymrl_per_sample: stratified draw (:228-239): segment = total/B,
 *     v = seg*b + seg*u[b] (u f64[B] in [0,1), or NULL -> Philox(seed, counter, b)),
 *     get_index descent (:130-144); idx_out data (A) or tree (B) indices, prio_out f64,
 *     w_out f32[B] = (size * p/total)^(-beta) / max — float32 normalisation for
 *     variant A (:226,:239,:241), float64 for variant B (ddqn_per_cartpole.py:135-138).
 */
size_t gymrl_per_workspace_bytes(int B);
int gymrl_per_update(double* tree, int64_t cap, const int32_t* idx, int64_t idx_start,
                     int idx_is_tree, const double* prio, const double* prio_scalar_dev,
                     double prio_scalar, int B, const int64_t* idx_start_dev, void* workspace, void* stream);
int gymrl_per_max_leaf(const double* tree, int64_t cap, double* out, void* workspace,
                       void* stream);
int gymrl_per_priorities(const float* td, int B, double alpha, double eps, double clip,
                         double* prio_out, void* stream);
/* update_priorities of a sampled batch (rainbow_dqn_cartpole.py:258-261) in two launches: gymrl_per_priorities' transform of
 * td inside gymrl_per_update's leaf pass (data indices idx i32[B], B <= 512, cap < 2^30), and — max_out != NULL — the maximum
 * over the leaves AFTER the update (gymrl_per_max_leaf: what the next store gives its new rows) computed by extra workgroups
 * of the ancestor launch; `ticket` u32[1] must be zero before the first call and is left zero.  The tree and max_out hold the
 * bits of gymrl_per_priorities -> gymrl_per_update -> gymrl_per_max_leaf. */
int gymrl_per_update_td(double* tree, int64_t cap, const int32_t* idx, const float* td, int B, double alpha, double eps,
                        double clip, double* max_out, unsigned int* ticket, void* workspace, void* stream);
int gymrl_per_sample(const double* tree, int64_t cap, const double* u, uint64_t seed,
                     uint64_t counter, int B, int64_t size, double beta, int variant_b,
                     int32_t* idx_out, double* prio_out, float* w_out, const void* dev, void* workspace,
                     void* stream);

/* R1: NoisyLinear.scale_noise + reset_noise — rainbow_dqn_cartpole.py:77-87.
 * f(x) = sign(x)*sqrt(|x|); w_eps[out,in] = outer(f(eps_out), f(eps_in)); b_eps = f(eps_out).
 * eps_in/eps_out raw N(0,1) f32 (parity mode) or NULL -> Box-Muller on Philox(seed, counter). */
int gymrl_noisy_noise(const float* eps_in_raw, const float* eps_out_raw, uint64_t seed,
                      uint64_t counter, int in_features, int out_features, float* w_eps_out,
                      float* b_eps_out, const uint64_t* counter_dev, void* stream);

/* D3: epsilon-greedy action selection for N envs — dqn_cartpole.py:117-133.
 * u f32[N,2] uniforms (or NULL -> Philox): u[.,0] < eps -> action = floor(u[.,1]*A) else argmax q. */
int gymrl_epsilon_greedy(const float* q, const float* u, uint64_t seed, uint64_t counter,
                         int64_t env_id0, int n, int A, float epsilon, int32_t* act_out,
                         void* stream);

/*
 * D4 / R4: TD target + loss forward/backward on the Q heads —
 * dqn_cartpole.py:157-161 (q_next_online NULL: max_a' Q_tgt), rainbow_dqn_cartpole.py:319-338
 * and ddqn_per_cartpole.py:224-233 (double DQN: a* = argmax q_next_online, weights w).
 *   y = r + gamma_n * q_tgt(s', a*) * (1 - flag);  td = q(s,a) - y
 *   loss = mean(td^2 * w) (w NULL -> 1, i.e. F.mse_loss);  dq[b,a] = 2*td*w/B
 * td_out f32[B] (feeds update_priorities), dq_out f32[B,A], loss_sum f64[1] += sum td^2*w.
 */
int gymrl_dqn_td_loss(const float* q, const float* q_next_online, const float* q_next_target,
                      const int32_t* act, const float* rew, const float* flag, const float* w,
                      int B, int A, double gamma_n, float* td_out, float* dq_out,
                      double* loss_sum, void* workspace, void* stream);

/*
 * A1: Actor.sample — sac_pendulum.py:76-87 — forward and backward.
 *   x = mean + exp(log_std)*eps; a = tanh(x)*bound;
 *   logp = sum_j [ Normal(mean,std).log_prob(x) - log(bound*(1 - tanh(x)^2) + 1e-6) ]
 * fwd: mean, log_std, eps f32[B,A] -> action f32[B,A], logp f32[B].
 * bwd: d_action f32[B,A] (may be NULL), d_logp f32[B] -> d_mean, d_log_std f32[B,A].
 */
int gymrl_sac_sample_fwd(const float* mean, const float* log_std, const float* eps, int B, int A,
                         float bound, float* action_out, float* logp_out, void* stream);
int gymrl_sac_sample_bwd(const float* mean, const float* log_std, const float* eps,
                         const float* d_action, const float* d_logp, int B, int A, float bound,
                         float* d_mean_out, float* d_log_std_out, void* stream);

/*
 * A4: SACTrainer.update pieces — sac_pendulum.py:233-263.
 *   target:  y = r + gamma*(1-done)*(min(q1n,q2n) - alpha*logp_n)            (:233-237)
 *   critic:  loss = mse(q1,y) + mse(q2,y); dq1 = 2(q1-y)/B, dq2 = 2(q2-y)/B   (:239-242)
 *   actor:   loss = mean(alpha*logp - min(q1,q2)); dlogp = alpha/B,
 *            dq_k = -(1/B) on the smaller critic (tie 1/2,1/2)                (:248-251)
 *   alpha:   loss = -mean(log_alpha*(logp + target_entropy)); one Adam step on the
 *            float64 scalar log_alpha (state m, v f64[1])                     (:257-263)
 * log_alpha f64[1] lives on the device; alpha = (float)exp(log_alpha) inside the kernels.
 * sums f64[4] += (critic loss sum, actor loss sum, sum(logp + target_entropy), 0).
 * alpha steps: bias_dev f64[2] or NULL = device-resident {1 - beta1^step, 1 - beta2^step}; with it `step`
 * is ignored and a captured hipGraph of the update replays unchanged (see gymrl_adam_step).
 */
int gymrl_sac_target(const float* rew, const float* done, const float* q1n, const float* q2n,
                     const float* logp_n, const double* log_alpha, int B, double gamma,
                     float* y_out, void* stream);
int gymrl_sac_critic_loss(const float* q1, const float* q2, const float* y, int B,
                          float* dq1_out, float* dq2_out, double* sums, void* workspace,
                          void* stream);
int gymrl_sac_actor_loss(const float* logp, const float* q1, const float* q2,
                         const double* log_alpha, int B, double target_entropy,
                         float* dlogp_out, float* dq1_out, float* dq2_out, double* sums,
                         void* workspace, void* stream);
int gymrl_sac_alpha_step(double* log_alpha, double* m, double* v, const double* sums, int B,
                         double lr, double beta1, double beta2, double eps, int64_t step,
                         const double* bias_dev, double* alpha_loss_out, void* stream);

/*
 * TD3 / DDPG — ddpg_pendulum.py:135-195, td3_pendulum.py:156-228.  Their Bellman target is
 * gymrl_sac_target with logp_n == 0 (r + gamma (1 - done) min(Q1', Q2'); DDPG passes Q' twice), TD3's critic
 * loss is gymrl_sac_critic_loss, the Polyak updates gymrl_soft_update; what is new:
 *   noisy_action : mode 0 = exploration noise of select_action (numpy float64: clip(mu + eps*std, +-bound),
 *                  :143-147 / :164-168), mode 1 = TD3 target-policy smoothing (torch float32:
 *                  clamp(mu + clamp(eps*std, +-noise_clip), +-bound), :191-196).  eps f64[n] explicit N(0,1)
 *                  draws or NULL -> Box-Muller on Philox(seed, counter, element).
 *   mse_loss     : one critic's F.mse_loss(q, y): dq = 2 (q - y)/B, sum_out += sum (q - y)^2   (ddpg :178-179)
 *   neg_mean_loss: actor loss -mean(Q(s, mu(s))): dq = -1/B, sum_out += sum q                  (ddpg :185, td3 :213)
 * sum_out f64[1] is accumulated into (zero it first); workspace as for the SAC losses.
 */
int gymrl_noisy_action(const float* mu, const double* eps, uint64_t seed, uint64_t counter, int64_t n,
                       int mode, double std, double noise_clip, double bound, float* out, void* stream);
int gymrl_mse_loss(const float* q, const float* y, int B, float* dq_out, double* sum_out, void* workspace,
                   void* stream);
int gymrl_neg_mean_loss(const float* q, int B, float* dq_out, double* sum_out, void* workspace, void* stream);

/*
 * Discrete SAC — sac_cartpole.py:148-227: expectation over the A <= 8 actions, float32 throughout
 * (log_alpha is a float32 scalar there).  probs = the actor's softmax output [B, A]; q* [B, A].
 *   dsac_target      : y = r + gamma (1 - done) (sum_a p'(a) min(Q1', Q2')(a) + alpha H(p')), log p = log(p + 1e-8)   :171-181
 *   dsac_critic_loss : F.mse_loss(q.gather(1, a), y) for both critics; dq [B, A] non-zero at the taken action;
 *                      sums f64[2] += (sum e1^2, sum e2^2)                                                         :183-186
 *   dsac_actor_loss  : L = mean(-alpha H(p) - sum_a p(a) min(Q1, Q2)(a)); dprobs = dL/dp (backprop through the
 *                      softmax by the caller); sums f64[2] += (sum (-alpha H - min_q), sum H)                      :196-203
 *   dsac_alpha_step  : L_alpha = exp(log_alpha) mean(H - H_target) from sums[1]; one float32 Adam step on log_alpha :209-215
 */
int gymrl_dsac_target(const float* probs_n, const float* q1n, const float* q2n, const float* rew,
                      const float* done, const float* log_alpha, int B, int A, double gamma, float* y_out,
                      void* stream);
int gymrl_dsac_critic_loss(const float* q1, const float* q2, const int32_t* act, const float* y, int B, int A,
                           float* dq1_out, float* dq2_out, double* sums, void* workspace, void* stream);
int gymrl_dsac_actor_loss(const float* probs, const float* q1, const float* q2, const float* log_alpha, int B,
                          int A, float* dprobs_out, double* sums, void* workspace, void* stream);
int gymrl_dsac_alpha_step(float* log_alpha, float* m, float* v, const double* sums, int B,
                          double target_entropy, double lr, double beta1, double beta2, double eps,
                          int64_t step, const double* bias_dev, double* alpha_loss_out, void* stream);

/*
 * N1-N3: utils/normalization.py — RunningMeanStd.update :12-22 (Welford, population
 * std, n == 1 sets std = x), Normalization.__call__ :29-35, RewardScaling :38-52.
 * stats f64[2 + 3*D] = (n, unused, mean[D] (float32 values), S[D], std[D]).
 * x f32[N,D] is consumed in env order 0..N-1 exactly like N successive reference calls
 * (the reference has ONE stream; this is its sequential-equivalent batch semantics).
 *   running_norm:   update (if update != 0) then y = (x - mean)/(std + 1e-8)
 *   reward_scaling: R[env] = gamma*R[env] + r[env]; update(R); y = r/(std + 1e-8);
 *                   R[env] := 0 where done (RewardScaling.reset at episode start)
 */
int gymrl_running_norm(const float* x, int N, int D, double* stats, int update, float* y_out,
                       void* stream);
int gymrl_reward_scaling(const float* r, const uint8_t* done, int N, double gamma, double* R,
                         double* stats, float* y_out, void* stream);

/* ============================================ fused off-policy vector step ===== */
/*
 * One SAC vector step of sac_pendulum.py:269-310 as ONE launch (gymrl_sac_step) or five (gymrl_sac_act_step + gymrl_sac_update's
 * four) instead of ~60 (csrc/offpolicy_step.hip; round 3's step:
 * 3 gymrl_lin_fwd + sample + env step + append + index draw + gather + ~45 layer / loss / optimiser launches of 4-14 us).
 * A forward or input-gradient pass never mixes batch rows, so ONE workgroup carries a 16-row slab of the batch through a
 * whole chain of layers in LDS (tile bodies of csrc/lin_device.hpp: the very MFMA sequence of gymrl_lin_*), and only the
 * weight gradients, which reduce over the batch, need a kernel boundary:
 *   gymrl_sac_act_step   acting (:278-283): Actor forward on the N observations, reparameterised draw, Pendulum step with
 *                        auto-reset, replay row (obs, action, reward, TERMINAL next obs, done) at (cursor + env) % cap
 *   gymrl_sac_update     update() (:213-267), four launches:
 *     P1 rows   index draw (gymrl_uniform_indices' permutation) + ring gather; a', logp' = Actor.sample(s'); target
 *               Q(s', a'); y; Q(s, a); critic loss gradient; input-gradient chain of both Q networks
 *     P2 tiles  every critic weight / bias gradient tile (gymrl_lin_bwd_weight's order) + Adam on that tile + the soft target
 *               update of the element just written (gymrl_adam_step's expressions); critic loss sum
 *     P3 rows   a, logp = Actor.sample(s); Q(s, a) of the UPDATED critic; actor loss gradient; the chain back through both
 *               Q networks to the action, through the sample, through the actor
 *     P4 tiles  actor weight gradients + Adam; actor loss / temperature sums; the float64 temperature step
 *   gymrl_sac_step       gymrl_sac_act_step followed by gymrl_sac_update in ONE grid: the five phases are block ranges, a phase
 *                        that needs an earlier one complete waits on that phase's counter in the workspace (release / acquire at
 *                        device scope) instead of a launch boundary, and does before the wait what does not depend on it (P1's index
 *                        draw, P3's loads).  The blocks that can wait are fewer than the compute units, so the blocks they wait
 *                        for always get one; the last block to finish zeroes the counters.  Same arguments, same results.
 * Results are those of the layer-by-layer path bit for bit (same products, same orders; tests/test_fused_step_gpu.py), which
 * tests/test_trainers_gpu.py pins against the reference.  Limits: H % 4 == 0, H <= 256, D <= 8, A <= 4, B <= 8192
 * (gymrl_sac_step: B <= 256; -22 otherwise: use the layer-by-layer path).  Above 256 rows the row kernels run on 1-D grids with a
 * slab's workgroups adjacent in dispatch order, above 512 the weight gradients are two launches (slice partials in
 * gymrl_lin_bwd_weight's cut, then their ordered sums + the optimiser step): still the layer path's bits.  Pointers are device pointers; W = [out][in] row-major (nn.Linear).
 */
typedef struct { float* w[4]; float* b[4]; } gymrl_sac_actor_params;     /* fc1, fc2, mean, log_std (sac_pendulum.py:58-61) */
typedef struct { float* w[6]; float* b[6]; } gymrl_sac_critic_params;    /* fc1..fc6: fc1-3 = Q1, fc4-6 = Q2 (:108-113) */
typedef struct {
  int N, D, A, H;                          /* envs, obs dim, action dim, hidden width */
  int env_kind;                            /* GYMRL_ENV_PENDULUM */
  void* env_state; uint64_t env_seed; int64_t env_id0;
  const float* obs;                        /* f32[N, D] the observations to act on */
  float* obs_out;                          /* f32[N, D] next observations (post-reset where an episode ended) */
  const float* eps;                        /* f32[N, A] N(0,1) draws, or NULL: Philox (noise_seed, noise_counter, stream 2, env * A + j) */
  uint64_t noise_seed, noise_counter; const uint64_t* noise_counter_dev;
  float bound, log_std_min, log_std_max;
  gymrl_sac_actor_params actor;
  /* replay ring (gymrl_replay_append's layout); cursor_dev int64[1] or NULL */
  float* r_state; uint32_t* r_action; float* r_reward; float* r_next; uint8_t* r_flag; int64_t cap, cursor;
  const int64_t* cursor_dev;
  /* per-step outputs of the episode bookkeeping (any may be NULL) */
  float* action_out; float* rew_out; uint8_t* done_out; float* ep_ret_out; double* ep_stats;
  const float* images;                     /* gymrl_sac_update_args.images of the same trainer, or NULL (reads actor.fc2 in place) */
} gymrl_sac_act_args;
typedef struct {
  int B, D, A, H;
  float gamma, bound, log_std_min, log_std_max, target_entropy;
  double tau;                              /* soft target update rate (a python float: rounded like gymrl_adam_step does) */
  const float* r_state; const uint32_t* r_action; const float* r_reward; const float* r_next; const uint8_t* r_flag;
  const int32_t* idx;                      /* i32[B] explicit rows, or NULL: the keyed permutation of gymrl_uniform_indices */
  uint64_t idx_seed, idx_counter; int64_t idx_size; const void* idx_dev;     /* idx_dev: {uint64 counter; int64 size} */
  const float* eps_next; const float* eps_cur;     /* f32[B, A] each, or NULL: Philox streams 3 / 4 of (noise_seed, noise_counter) */
  uint64_t noise_seed, noise_counter; const uint64_t* noise_counter_dev;
  gymrl_sac_actor_params actor; gymrl_sac_critic_params critic, target;
  /* flat parameter buffers and their Adam moments (same layout): m of a parameter p lives at critic_m + (p - critic_p) */
  float* actor_p; float* actor_m; float* actor_v; float* critic_p; float* critic_m; float* critic_v;
  float adam_critic[4], adam_actor[4];     /* gymrl_adam_bias' block {lr/(1-b1^t), 1/(1-b1^t), sqrt(1-b2^t), 0} */
  const float* adam_critic_dev; const float* adam_actor_dev;      /* the same from device memory (hipGraph replay) or NULL */
  double beta1, beta2, eps_adam;
  double* log_alpha; double* alpha_m; double* alpha_v; double lr_alpha;
  double alpha_bias[2]; const double* alpha_bias_dev;             /* {1 - 0.9^t, 1 - 0.999^t} (gymrl_sac_alpha_step) */
  double* sums;                            /* f64[3] out: critic loss sum, actor loss sum, sum of (logp + target_entropy) */
  double* alpha_loss;                      /* f64[1] out or NULL */
  void* workspace;                         /* >= gymrl_sac_update_workspace_bytes(B, D, A, H); ZEROED once before the first call (the
                                            * hand-off flags between the row phases' workgroups and gymrl_sac_step's phase counters
                                            * live in it and are left zero) */
  /* Weight images of the H x H layers (H % 16 == 0), f32[8][H*H], or NULL (every layer read in place).  The MFMA operands of
   * actor.fc2, critic.fc2 / fc5, target.fc2 / fc5 (forward) and actor.fc2, critic.fc2 / fc5 (input gradient) as contiguous
   * 1 KiB blocks per wave-wide load (csrc/lin_device.hpp): a slab kernel streams a whole weight matrix through one compute
   * unit, 3x faster from the image.  gymrl_sac_update keeps the images equal to the parameters it updates;
   * gymrl_sac_pack_images rebuilds them after anything else wrote the parameters (load_state_dict, a checkpoint, ...). */
  float* images;
} gymrl_sac_update_args;
size_t gymrl_sac_update_workspace_bytes(int B, int D, int A, int H);
int gymrl_sac_pack_images(const gymrl_sac_update_args* args, void* stream);
size_t gymrl_sac_args_bytes(int which);      /* sizeof(gymrl_sac_act_args) (0) / sizeof(gymrl_sac_update_args) (1): a binding checks its mirror */
int gymrl_sac_act_step(const gymrl_sac_act_args* args, void* stream);
int gymrl_sac_update(const gymrl_sac_update_args* args, void* stream);
int gymrl_sac_step(const gymrl_sac_act_args* act, const gymrl_sac_update_args* update, void* stream);   /* both, one launch */

/*
 * The same treatment for Rainbow's vector step (rainbow_dqn_cartpole.py:363-405; csrc/offpolicy_step.hip).  The NoisyLinear
 * bookkeeping (which draw a forward uses, where its epsilons live) stays with gymrl_noisy_combine / gymrl_noisy_split, the sum
 * tree with gymrl_per_*; the Linear / loss / env / n-step launches between them — ~20 of a step's ~45 — become three:
 *   gymrl_rainbow_act_step     greedy acting on the noisy Q (:293-309, :371): fc1, fc2, the stacked noisy heads with the dueling
 *                              combination, argmax (first maximum), CartPole step with auto-reset, `terminal = done and step !=
 *                              max_steps - 1` (:376), the n-step window push and the emitted ring row (gymrl_nstep_push)
 *   gymrl_rainbow_update       rows: ring gather at the sampled indices; policy(s') | target(s') | policy(s) through fc1, fc2 and
 *                              their stacked heads (W_heads [3][A+1][H], b_heads [3][A+1] from gymrl_noisy_combine: first draw,
 *                              target means, second draw); double-DQN target, IS-weighted TD loss gradient (gymrl_dqn_td_loss's
 *                              expressions), dueling backward, input-gradient chain;  tiles: dW / db of the stacked head (for
 *                              gymrl_noisy_split), fc2 and fc1 written where the caller says (the flat gradient buffer's views:
 *                              clip_grad_norm_ needs every gradient before Adam), loss sum
 * Bit-identical to the layer-by-layer path (same tile bodies, same scalar expressions).  Limits: H % 4 == 0, H <= 256, D <= 8,
 * A <= 3, B <= 8192 (above 256 / 512 rows: as gymrl_sac_update); CartPole-v1 for the act step.
 */
typedef struct {
  int N, D, A, H;
  int env_kind;                              /* GYMRL_ENV_CARTPOLE */
  void* env_state; uint64_t env_seed; int64_t env_id0;
  const float* obs; float* obs_out;          /* f32[N, D] in / next observations out */
  const float* fc1_w; const float* fc1_b; const float* fc2_w; const float* fc2_b;
  const float* head_w; const float* head_b;  /* stacked effective head [A+1][H], [A+1] (advantage rows, then value) */
  int max_episode_steps;
  /* n-step windows + ring (gymrl_nstep_push's arguments) */
  float* w_state; int32_t* w_action; float* w_reward; float* w_next; uint8_t* w_terminal; uint8_t* w_done;
  int n_steps; int64_t pushes; double gamma;
  float* r_state; uint32_t* r_action; float* r_reward; float* r_next; uint8_t* r_flag; int64_t cap, cursor;
  const int64_t* push_dev;                   /* {pushes, cursor} from the device (hipGraph replay) or NULL */
  int32_t* action_out; float* rew_out; uint8_t* done_out; float* ep_ret_out; double* ep_stats;   /* any may be NULL */
  const float* fc2_img;                      /* forward weight image of fc2_w (gymrl_weight_image; H % 16 == 0) or NULL: read in place */
} gymrl_rainbow_act_args;
typedef struct {
  int B, D, A, H;
  float gamma_n;                             /* gamma ** n_steps */
  const float* r_state; const uint32_t* r_action; const float* r_reward; const float* r_next; const uint8_t* r_flag;
  const int32_t* idx; const float* is_weight;        /* i32[B] sampled rows, f32[B] importance weights (NULL: 1) */
  const float* p_fc1_w; const float* p_fc1_b; const float* p_fc2_w; const float* p_fc2_b;    /* policy network */
  const float* t_fc1_w; const float* t_fc1_b; const float* t_fc2_w; const float* t_fc2_b;    /* target network */
  const float* head_w; const float* head_b;  /* [3][A+1][H], [3][A+1]: policy on s' (first draw), target on s', policy on s (second draw) */
  float* td_out;                             /* f32[B] TD errors (update_priorities) */
  double* loss_sum;                          /* f64[1] out: sum of w * td^2 */
  float* d_fc1_w; float* d_fc1_b; float* d_fc2_w; float* d_fc2_b; float* d_head_w; float* d_head_b;   /* gradients (overwritten) */
  void* workspace;                           /* >= gymrl_rainbow_update_workspace_bytes(B, D, A, H); ZEROED once before the first call
                                              * (the hand-off flags between the row phase's three workgroups per slab live in it) */
  /* gymrl_noisy_split inside the weight-gradient launch (split_heads != 0; d_head_w / d_head_b are then not written): the stacked
   * head's gradient goes straight to the two NoisyLinear layers' parameters, [0] = advantage (rows 0 .. A-1), [1] = value (row A):
   * d mu = dW, d sigma = dW * eps with the SECOND draw's epsilons (rainbow_dqn_cartpole.py:92-93 under autograd) */
  int split_heads;
  float* dw_mu[2]; float* dw_sigma[2]; float* db_mu[2]; float* db_sigma[2]; const float* w_eps[2]; const float* b_eps[2];
  /* Weight images of the H x H layer (H % 16 == 0; gymrl_weight_image) or NULL (read in place): p_fc2_w forward and
   * input-gradient, t_fc2_w forward.  The CALLER keeps them equal to the parameters: the Rainbow trainer rebuilds all three in
   * the step's gymrl_noisy_combine_images launch, between the optimiser step and the next acting forward. */
  const float* p_fc2_img_f; const float* p_fc2_img_b; const float* t_fc2_img_f;
} gymrl_rainbow_update_args;
size_t gymrl_rainbow_update_workspace_bytes(int B, int D, int A, int H);
size_t gymrl_rainbow_args_bytes(int which);  /* sizeof(gymrl_rainbow_act_args) (0) / sizeof(gymrl_rainbow_update_args) (1) */
int gymrl_rainbow_act_step(const gymrl_rainbow_act_args* args, void* stream);
int gymrl_rainbow_update(const gymrl_rainbow_update_args* args, int phase, void* stream);   /* phase 0: both launches; 1: rows (td_out is complete after it: update_priorities may start); 2: tiles */

#ifdef __cplusplus
}
#endif
#endif /* GYMRL_H */
