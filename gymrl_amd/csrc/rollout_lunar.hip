// rollout_lunar.hip — PPO's collect_rollout (ppo_lunarlander.py:198-231) for LunarLander as ONE
// persistent launch per chunk of vector steps.
//
// Launched step by step, a vector step is `lunar_step_kernel` (one wave per 16 envs, 256 waves at
// N = 4096, every wave on its own SIMD) + the policy forward + the sample kernel, and each launch
// lasts as long as its SLOWEST wave: with 4096 envs some wave always holds the worst contact
// configuration (two manifolds x two points on a body), so every step costs ~350 us although the
// average wave needs ~150 us (`tools/micro_lunar.py 16`).  Nothing couples two workgroups inside
// a rollout — the policy weights are frozen, envs are independent — so here a workgroup owns its
// 16 envs for `nsteps` steps and never waits for another one: the rollout costs the mean wave time
// per step instead of the max.  Per step, inside the workgroup:
//
//   all 4 waves  policy forward of the 16 observations (mlp_device.hpp: f32 MFMA, activations and
//                now also logits/value stay in LDS)
//   wave 0       folds step t-1 into its GAE chunk map (gae_online_compose), draws the action
//                (policy_device.hpp: same Philox keys / explicit noise as gymrl_categorical_sample),
//                writes act/logp/val[t], runs the Box2D step of its 16 envs four lanes per env
//                (env_lunar_device.hpp, inline reset pass where an episode ended), writes
//                rew/done/ep_ret[t] and obs[t+1] to the slab and the next observation tile
//                straight into the forward's LDS input
//
// Every arithmetic piece is the same device function the step-by-step path launches, so the slab
// is bit-identical to `collect_rollout()` without this kernel (tests/test_hip_parity.py).
#include "env_lunar_device.hpp"
#include "mlp_device.hpp"
#include "policy_device.hpp"
#include "mhc_policy_device.hpp"

using namespace gymrl;
using namespace gymrl::lunar;
namespace M = gymrl::mlp;

namespace {

constexpr int kThreads = M::kWaves * 64;
constexpr int kActions = 4, kObs = 8;
constexpr int kGaeChunk = 16;        // == gymrl_gae_chunk() (gae.hip kBlkTC)
constexpr int kDynFloats = M::kBufs * M::kRows * M::kStride + M::kRows * M::kInStride + M::kRows * M::kHeadStride;
constexpr int kDynBytes = 96 * 1024; // > 80 KB: one workgroup per CU (as mlp_forward_kernel)
constexpr int kDynBytesMhc = 32 * 1024;   // the mHC kernel's static LDS (policy tile, solver columns, parking) is already 64 KB: > 80 KB in all
static_assert(kDynFloats * 4 <= kDynBytes, "LDS carve-up");

__global__ __launch_bounds__(kThreads) void rollout_lunar_kernel(gymrl_rollout_lunar_args a, gymrl_mlp_desc d) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  __shared__ uint32_t lds_words[kLdsWords * kEnvBlock];            // the solver's per-lane columns (wave 0)
  __shared__ uint32_t lds_words_refill[kLdsWords * kEnvBlock];     // the same for the refill wave (wave 1)
  __shared__ uint32_t lds_park[kParkWords * kEnvBlock];            // wave 0's worlds between the steps of this launch
  float (*lds)[M::kRows * M::kStride] = reinterpret_cast<float (*)[M::kRows * M::kStride]>(dyn_lds);
  float* xin = dyn_lds + M::kBufs * M::kRows * M::kStride;
  float* head = xin + M::kRows * M::kInStride;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int N = a.n_envs, T = a.T;
  const int m0 = blockIdx.x * M::kRows;
  const int t_end = a.t0 + a.nsteps;
  // wave 0's view: four lanes per env
  const int role = lane & 3, row = lane >> 2, i = m0 + row;
  const bool valid = i < N;
  const LunarState st(a.env_state, N);
#ifdef GYMRL_LUNAR_PROF
  // probe layout of wg_ticks (u64): [2][G] start / end | [G][kProfSlots] per-workgroup sections | [3][N] per-env counters
  constexpr int kProfSlots = 24;
  unsigned long long* const wprof = a.wg_ticks ? reinterpret_cast<unsigned long long*>(a.wg_ticks) + 2 * gridDim.x + kProfSlots * blockIdx.x : nullptr;
  Lds slds_{lds_words + lane, wprof, lds_park + lane};
  if (a.wg_ticks && valid) { slds_.envp = reinterpret_cast<unsigned long long*>(a.wg_ticks) + (2 + kProfSlots) * gridDim.x + i; slds_.envn = N; }
  const Lds slds = slds_;
  if (wprof && lane == 0) {                       // where the hardware put this wave: HW_ID (wave / SIMD / pipe / CU / SH / SE) and XCC_ID
    wprof[20 + wave] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |
                       ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
  }
#else
  const Lds slds{lds_words + lane, lds_park + lane};
#endif
#ifdef GYMRL_LUNAR_PROF
  const Lds rlds{lds_words_refill + lane, nullptr};
#else
  const Lds rlds{lds_words_refill + lane};
#endif
  const double gl = (double)(float)(a.gamma * a.lam);              // NEP-50 float32 decay (see gae.hip)
  if (a.wg_ticks && tid == 0) a.wg_ticks[2 * blockIdx.x] = wall_clock64();   // profiling: 100 MHz ticks

  // observation tile of step t0 -> xin (zero padded to 64 columns once; later steps rewrite columns 0..7)
  for (int e = tid; e < M::kRows * M::kInStride; e += kThreads) xin[e] = 0.0f;
  __syncthreads();
  for (int e = tid; e < M::kRows * kObs; e += kThreads) {
    const int r = e / kObs, c = e - r * kObs;
    if (m0 + r < N) xin[r * M::kInStride + c] = a.obs[((size_t)a.t0 * N + m0 + r) * kObs + c];
  }
  const uint32_t vdefer = M::deferrable_stages(d, kActions);
  for (int t = a.t0; t <= t_end; ++t) {
    const bool tail = t == t_end;                   // only the bootstrap value of the finished rollout is left
    if (tail && t_end != T) break;
    LUNAR_PROF_MARK(rtb);
    __syncthreads();                                // xin of step t is complete
    LUNAR_PROF_MARK(rt0);
    if (wave == 0) LUNAR_PROF_ADD(slds, 16, rtb, rt0);  // wave 0 waiting for the refill / critic waves at the step's barrier
    // logits -> head[row][0..3] by all four waves; the critic's layers and the value head (-> head[row][4]) are not on the
    // way to the action: wave 2 runs them while wave 0 steps the envs (`vdefer` = their stage bits, 0 if the network has
    // no such split) and does what the value is needed for — val[t], the GAE delta of step t-1, the bootstrap value
    M::forward_tile(d, lds, xin, head, m0, N, tid, vdefer);
    LUNAR_PROF_MARK(rt1);
    if (wave == 0) LUNAR_PROF_ADD(slds, 8, rt0, rt1);   // policy forward
    auto value_part = [&]() {
      const float v = head[row * M::kHeadStride + kActions];
      if (valid && role == 0) {
        if (a.gae_running && t > 0) {               // V_t completes step t-1's delta
          const int tp = t - 1;
          const size_t o = (size_t)tp * N + i;
          double2* agg = reinterpret_cast<double2*>(a.gae_workspace) + (size_t)(tp / kGaeChunk) * N;
          gae_online_compose(a.rew[o], a.done[o], a.val[o], v, a.gamma, gl, (tp % kGaeChunk) == 0,
                             (tp % kGaeChunk) == kGaeChunk - 1 || tp == T - 1, a.gae_running, agg, N, i);
        }
        if (tail) {
          a.next_value[i] = v;
          // the rollout is complete and this lane has just closed its env's last chunk map: the blocked scan's carry pass for
          // this env here (128 maps this lane wrote itself), so that gymrl_gae (variant 3) is its apply launch alone
          if (a.gae_carry && a.gae_running) {
            const int C = (T + kGaeChunk - 1) / kGaeChunk;
            const double2* agg0 = reinterpret_cast<const double2*>(a.gae_workspace);
            // (scratch: the refill wave's solver columns — wave 1 does nothing at the tail, and the barrier at the top of this
            // iteration is behind its last use of them; element s of env `row` at [s][row])
            static_assert(sizeof(lds_words_refill) >= sizeof(double2) * kGaeCarrySeg * kEnvBlock / 4, "carry scratch");
            gae_carry_scan(agg0, C, N, i, reinterpret_cast<double*>(const_cast<double2*>(agg0) + (size_t)C * N),
                           reinterpret_cast<double2*>(lds_words_refill) + row, kEnvBlock / 4);
          }
        } else a.val[(size_t)t * N + i] = v;
      }
    };
    if (wave == 2 && vdefer) {
      M::forward_deferred(d, lds, head, m0, N, lane, vdefer);
      value_part();
#ifdef GYMRL_LUNAR_PROF
      if (wprof && lane == 0) wprof[17] += wall_clock64() - rt1;     // the critic wave's busy time
#endif
    }
    // wave 0 steps the envs; wave 1 (idle otherwise) keeps their next episodes prepared — ONE call site of the solver
    const bool refill_wave = wave == 1;
    if (wave == 0 || (refill_wave && !tail && a.refill)) {
      int act = 0;
      if (!refill_wave) {
        float z[kActions];
#pragma unroll
        for (int k = 0; k < kActions; ++k) z[k] = head[row * M::kHeadStride + k];
        if (!vdefer) value_part();
        if (!tail) {
          float lp, H;
          const size_t o = (size_t)t * N + (valid ? i : 0);
          act = categorical_pick<kActions>(z, a.noise_exp ? a.noise_exp + o * kActions : nullptr, a.seed,
                                           (uint64_t)(a.env_id0 + i), a.counter0 + (uint64_t)t, 0, lp, H);
          if (valid && role == 0) { a.act[o] = act; a.logp[o] = lp; }
        }
      }
      if (!tail) {
        const StepOut out{a.obs + (size_t)(t + 1) * N * kObs, nullptr, a.rew + (size_t)t * N, nullptr, nullptr,
                          a.done + (size_t)t * N, a.ep_ret ? a.ep_ret + (size_t)t * N : nullptr, nullptr, a.ep_stats};
        float o_next[8];
        LUNAR_PROF_MARK(rt2);
        if (!refill_wave) LUNAR_PROF_ADD(slds, 9, rt1, rt2);          // GAE compose + draw + slab writes
        // the worlds stay in LDS between the steps of this launch: HBM is read at its first step and written at its last
        const int io_mode = (t > a.t0 ? 1 : 0) | (t + 1 < t_end ? 2 : 0);
        lunar_step_quad(st, refill_wave ? rlds : slds, N, i, role, valid, act, a.seed, a.env_id0, out, o_next, refill_wave, io_mode);
        LUNAR_PROF_MARK(rt3);
        if (!refill_wave) LUNAR_PROF_ADD(slds, 10, rt2, rt3);         // whole env step (state load, world_step, reward, reset, store)
#ifdef GYMRL_LUNAR_PROF
        if (refill_wave && wprof && lane == 0) { wprof[18] += rt3 - rt2; wprof[19] += (rt3 - rt2) > 1000ull ? 1ull : 0ull; }   // the refill wave's busy time, steps on which it built a world (> 10 us)
#endif
        if (!refill_wave && valid && role < 2) {      // next policy input: straight into the forward's LDS tile
#pragma unroll
          for (int k = 0; k < 4; ++k) xin[row * M::kInStride + 4 * role + k] = o_next[4 * role + k];
        }
      }
    }
    if (tail) break;
  }
  if (a.wg_ticks && tid == 0) a.wg_ticks[2 * blockIdx.x + 1] = wall_clock64();
}

// The same rollout with PPO-full's mHC network as the policy (ppo_full_lunarlander.py collect_experience :440-505): the
// forward is mhc_policy_device.hpp's 16-row tile (the workgroup IS that kernel's workgroup), the behaviour policy's entropy is
// kept (`ent`, the entropy-ratio mask's reference :594), and with `gae_running2` both decoupled-lambda chunk maps are composed
// (G3: actor then critic, float64 decay factors, as gymrl_categorical_sample's online mode does step by step).
constexpr int kXStride = 16;
// its own function: the tile's ~170 registers and the solver's are allocated separately (inlined, the kernel spilled 396 B per lane)
__device__ __noinline__ void policy_tile_call(const mhc::PolicyArgs& p, mhc::PolicyLds& L, const float* obs_row, float* head) {
  mhc::policy_tile(p, L, obs_row, head, M::kHeadStride, head + kActions, M::kHeadStride, M::kRows);
}

__global__ __launch_bounds__(kThreads) void rollout_lunar_mhc_kernel(gymrl_rollout_lunar_args a, mhc::PolicyArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];  // (its size keeps one workgroup per CU)
  __shared__ mhc::PolicyLds L;
  __shared__ uint32_t lds_words[kLdsWords * kEnvBlock];            // the solver's per-lane columns (wave 0)
  __shared__ uint32_t lds_words_refill[kLdsWords * kEnvBlock];     // the same for the refill wave (wave 1)
  __shared__ uint32_t lds_park[kParkWords * kEnvBlock];            // wave 0's worlds between the steps of this launch
  float* xin = dyn_lds;                                            // [16][kXStride] observations
  float* head = xin + M::kRows * kXStride;                         // [16][kHeadStride]: logits 0..3, value 4
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int N = a.n_envs, T = a.T;
  const int m0 = blockIdx.x * M::kRows;
  const int t_end = a.t0 + a.nsteps;
  const int role = lane & 3, row = lane >> 2, i = m0 + row;        // wave 0's view: four lanes per env
  const bool valid = i < N;
  const int prow = 4 * wave + (lane >> 4);                         // the policy tile's view: this lane's row
  const LunarState st(a.env_state, N);
#ifdef GYMRL_LUNAR_PROF
  const Lds slds{lds_words + lane, nullptr, lds_park + lane}, rlds{lds_words_refill + lane, nullptr};
#else
  const Lds slds{lds_words + lane, lds_park + lane}, rlds{lds_words_refill + lane};
#endif
  const bool two = a.gae_running2 != nullptr;
  const double gl = two ? a.gamma * a.lam : (double)(float)(a.gamma * a.lam);
  const double gl2 = a.gamma * a.lam2;
  const size_t chunks = (size_t)((T + kGaeChunk - 1) / kGaeChunk);
  if (a.wg_ticks && tid == 0) a.wg_ticks[2 * blockIdx.x] = wall_clock64();
  for (int e = tid; e < M::kRows * kXStride; e += kThreads) xin[e] = 0.0f;
  __syncthreads();
  for (int e = tid; e < M::kRows * kObs; e += kThreads) {
    const int r = e / kObs, c = e - r * kObs;
    if (m0 + r < N) xin[r * kXStride + c] = a.obs[((size_t)a.t0 * N + m0 + r) * kObs + c];
  }
  for (int t = a.t0; t <= t_end; ++t) {
    const bool tail = t == t_end;                   // only the bootstrap value of the finished rollout is left
    if (tail && t_end != T) break;
    __syncthreads();                                // xin of step t is complete (and the previous step's head reads are done)
    policy_tile_call(p, L, xin + prow * kXStride, head);
    __syncthreads();                                // logits -> head[row][0..3], value -> head[row][4]
    const bool refill_wave = wave == 1;               // wave 1 keeps the next episodes prepared while wave 0 steps
    if (wave == 0 || (refill_wave && !tail && a.refill)) {
      int act = 0;
      if (!refill_wave) {
        float z[kActions];
#pragma unroll
        for (int k = 0; k < kActions; ++k) z[k] = head[row * M::kHeadStride + k];
        const float v = head[row * M::kHeadStride + kActions];
        if (valid && role == 0) {
          if (a.gae_running && t > 0) {               // V_t completes step t-1's delta
            const int tp = t - 1;
            const size_t o = (size_t)tp * N + i;
            double2* agg = reinterpret_cast<double2*>(a.gae_workspace) + (size_t)(tp / kGaeChunk) * N;
            const int first = (tp % kGaeChunk) == 0, last = (tp % kGaeChunk) == kGaeChunk - 1 || tp == T - 1;
            gae_online_compose(a.rew[o], a.done[o], a.val[o], v, a.gamma, gl, first, last, a.gae_running, agg, N, i);
            if (two) gae_online_compose(a.rew[o], a.done[o], a.val[o], v, a.gamma, gl2, first, last, a.gae_running2, agg + chunks * N, N, i);
          }
          if (tail) a.next_value[i] = v;
        }
        if (!tail) {
          float lp, H;
          const size_t o = (size_t)t * N + (valid ? i : 0);
          act = categorical_pick<kActions>(z, a.noise_exp ? a.noise_exp + o * kActions : nullptr, a.seed,
                                           (uint64_t)(a.env_id0 + i), a.counter0 + (uint64_t)t, 0, lp, H);
          if (valid && role == 0) {
            a.act[o] = act; a.logp[o] = lp; a.val[o] = v;
            if (a.ent) a.ent[o] = H;
          }
        }
      }
      if (!tail) {
        const StepOut out{a.obs + (size_t)(t + 1) * N * kObs, nullptr, a.rew + (size_t)t * N, nullptr, nullptr,
                          a.done + (size_t)t * N, a.ep_ret ? a.ep_ret + (size_t)t * N : nullptr, nullptr, a.ep_stats};
        float o_next[8];
        // the worlds stay in LDS between the steps of this launch: HBM is read at its first step and written at its last
        const int io_mode = (t > a.t0 ? 1 : 0) | (t + 1 < t_end ? 2 : 0);
        lunar_step_quad(st, refill_wave ? rlds : slds, N, i, role, valid, act, a.seed, a.env_id0, out, o_next, refill_wave, io_mode);
        if (!refill_wave && valid && role < 2) {      // next policy input: straight into the forward's LDS tile
#pragma unroll
          for (int k = 0; k < 4; ++k) xin[row * kXStride + 4 * role + k] = o_next[4 * role + k];
        }
      }
    }
    if (tail) break;
  }
  if (a.wg_ticks && tid == 0) a.wg_ticks[2 * blockIdx.x + 1] = wall_clock64();
}

}  // namespace

extern "C" {

int gymrl_rollout_lunar(const gymrl_rollout_lunar_args* a, const gymrl_mlp_desc* policy, void* stream) {
  if (!a || !policy || !a->env_state || !a->obs || !a->act || !a->logp || !a->val || !a->rew || !a->done ||
      !a->next_value || a->n_envs <= 0 || a->T <= 0 || a->t0 < 0 || a->nsteps < 0 || a->t0 + a->nsteps > a->T)
    return -22;
  if (a->gae_running && !a->gae_workspace) return -22;
  // the policy must be a 2-output network on the LunarLander observation: logits [4] then value [1]
  int outs = 0, cols = 0;
  if (policy->n_stages <= 0 || policy->n_stages > GYMRL_MLP_MAX_STAGES) return -22;
  for (int s = 0; s < policy->n_stages; ++s) {
    const gymrl_mlp_stage& st = policy->stage[s];
    if (!st.W || st.in_dim <= 0 || st.out_dim <= 0 || st.in_dim > GYMRL_MLP_MAX_WIDTH || st.src < -1 || st.src > 2 ||
        st.dst < -1 || st.dst > 2 || (st.dst >= 0 && (st.dst == st.src || st.out_dim > GYMRL_MLP_MAX_WIDTH)) ||
        (st.src < 0 && st.in_dim != kObs) || st.act < GYMRL_ACT_NONE || st.act > GYMRL_ACT_RELU ||
        (reinterpret_cast<uintptr_t>(st.W) & 15))
      return -22;
    if (st.dst < 0) {
      if ((outs == 0 && st.out_dim != kActions) || (outs == 1 && st.out_dim != 1) || outs > 1) return -22;
      ++outs; cols += st.out_dim;
    }
  }
  if (outs != 2 || cols > M::kHeadStride) return -22;
  if (a->nsteps == 0 && a->t0 != a->T) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rollout_lunar_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDynBytes) !=
        hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr_set = true;
  }
  const int blocks = (a->n_envs + M::kRows - 1) / M::kRows;
  hipLaunchKernelGGL(rollout_lunar_kernel, dim3(blocks), dim3(kThreads), kDynBytes, (hipStream_t)stream, *a, *policy);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

#ifdef GYMRL_LUNAR_PROF
int gymrl_mlp_fwd_prof_read(unsigned long long* out16, int reset) {      // probe build only (not in include/gymrl.h)
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(M::g_fwd_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(M::g_fwd_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

#ifdef GYMRL_LUNAR_PROF
int gymrl_mhc_policy_prof_read(unsigned long long* out16, int reset) {   // probe build only (not in include/gymrl.h)
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mhc::g_pol_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mhc::g_pol_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

int gymrl_rollout_lunar_mhc(const gymrl_rollout_lunar_args* a, const gymrl_mhc_policy* policy, void* stream) {
  if (!a || !policy || !a->env_state || !a->obs || !a->act || !a->logp || !a->val || !a->rew || !a->done ||
      !a->next_value || a->n_envs <= 0 || a->T <= 0 || a->t0 < 0 || a->nsteps < 0 || a->t0 + a->nsteps > a->T)
    return -22;
  if (a->gae_running && !a->gae_workspace) return -22;
  if (a->gae_running2 && (!a->gae_running || !(a->lam2 > 0.0))) return -22;
  mhc::PolicyArgs p{};
  if (const int rc = mhc::policy_fill(p, policy)) return rc;
  if (p.obs_dim != kObs || p.n_act != kActions) return -22;       // LunarLander: 8 observations, 4 actions
  if (a->nsteps == 0 && a->t0 != a->T) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rollout_lunar_mhc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDynBytesMhc) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr_set = true;
  }
  const int blocks = (a->n_envs + M::kRows - 1) / M::kRows;
  hipLaunchKernelGGL(rollout_lunar_mhc_kernel, dim3(blocks), dim3(kThreads), kDynBytesMhc, (hipStream_t)stream, *a, p);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
