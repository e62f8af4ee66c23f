// rollout_classic.hip — PPO's collect_rollout (ppo_lunarlander.py:198-231) for CartPole-v1 as ONE persistent launch per
// chunk of vector steps: rollout_lunar.hip's scheme with the classic-control stepper in place of the Box2D one.
//
// Step by step, a CartPole vector step is the policy forward + the sample kernel + cartpole_step_kernel: three launches
// of a few microseconds of work each, i.e. the launch floor.  Nothing couples two workgroups inside a rollout (frozen
// weights, independent envs), so a workgroup owns its 16 envs for `nsteps` steps: all four waves run the policy forward of
// the 16 observations (mlp_device.hpp: f32 MFMA, logits and value stay in LDS), then lanes 0..15 of wave 0 fold step
// t-1 into its GAE chunk map, draw the action (policy_device.hpp: the Philox keys / explicit noise of
// gymrl_categorical_sample), step their env (env_classic_device.hpp: the arithmetic of gymrl_env_step), write the slab
// rows and put the next observation into the forward's LDS input.  Every arithmetic piece is the device function the
// step-by-step path launches: the slab is bit-identical to collect_rollout() without this kernel
// (tests/test_hip_parity.py::test_persistent_rollout_cartpole_bit_identical_to_stepwise).
#include "env_classic_device.hpp"
#include "mlp_device.hpp"
#include "policy_device.hpp"

using namespace gymrl;
namespace M = gymrl::mlp;

namespace {

constexpr int kThreads = M::kWaves * 64;
constexpr int kActions = 2, kObs = 4;
constexpr int kGaeChunk = 16;        // == gymrl_gae_chunk() (gae.hip kBlkTC)
constexpr int kDynFloats = M::kBufs * M::kRows * M::kStride + M::kRows * M::kInStride + M::kRows * M::kHeadStride;
constexpr int kDynBytes = 96 * 1024; // > 80 KB: one workgroup per CU (as mlp_forward_kernel)
static_assert(kDynFloats * 4 <= kDynBytes, "LDS carve-up");

__global__ __launch_bounds__(kThreads) void rollout_cartpole_kernel(gymrl_rollout_lunar_args a, gymrl_mlp_desc d) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  float (*lds)[M::kRows * M::kStride] = reinterpret_cast<float (*)[M::kRows * M::kStride]>(dyn_lds);
  float* xin = dyn_lds + M::kBufs * M::kRows * M::kStride;
  float* head = xin + M::kRows * M::kInStride;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int N = a.n_envs, T = a.T;
  const int m0 = blockIdx.x * M::kRows;
  const int t_end = a.t0 + a.nsteps;
  const int row = lane, i = m0 + row;                              // wave 0's view: one lane per env (lanes 0..15)
  const bool valid = row < M::kRows && i < N;
  const CartPoleState st(a.env_state, N);
  const double gl = (double)(float)(a.gamma * a.lam);              // NEP-50 float32 decay (see gae.hip)

  for (int e = tid; e < M::kRows * M::kInStride; e += kThreads) xin[e] = 0.0f;
  __syncthreads();
  for (int e = tid; e < M::kRows * kObs; e += kThreads) {
    const int r = e / kObs, c = e - r * kObs;
    if (m0 + r < N) xin[r * M::kInStride + c] = a.obs[((size_t)a.t0 * N + m0 + r) * kObs + c];
  }
  for (int t = a.t0; t <= t_end; ++t) {
    const bool tail = t == t_end;                   // only the bootstrap value of the finished rollout is left
    if (tail && t_end != T) break;
    __syncthreads();                                // xin of step t is complete
    M::forward_tile(d, lds, xin, head, m0, N, tid, 0u);
    __syncthreads();                                // logits / value of the 16 rows are in `head`
    if (wave == 0) {
      ClassicStep<4> r;
      r.done = false; r.ret = 0.0; r.len = 0;
      if (valid) {
        const float v = head[row * M::kHeadStride + kActions];
        if (a.gae_running && t > 0) {               // V_t completes step t-1's delta
          const int tp = t - 1;
          const size_t o = (size_t)tp * N + i;
          double2* agg = reinterpret_cast<double2*>(a.gae_workspace) + (size_t)(tp / kGaeChunk) * N;
          gae_online_compose(a.rew[o], a.done[o], a.val[o], v, a.gamma, gl, (tp % kGaeChunk) == 0,
                             (tp % kGaeChunk) == kGaeChunk - 1 || tp == T - 1, a.gae_running, agg, N, i);
        }
        if (tail) a.next_value[i] = v;
        else {
          const size_t o = (size_t)t * N + i;
          a.val[o] = v;
          float z[kActions], lp, H;
#pragma unroll
          for (int k = 0; k < kActions; ++k) z[k] = head[row * M::kHeadStride + k];
          const int act = categorical_pick<kActions>(z, a.noise_exp ? a.noise_exp + o * kActions : nullptr, a.seed,
                                                     (uint64_t)(a.env_id0 + i), a.counter0 + (uint64_t)t, 0, lp, H);
          a.act[o] = act; a.logp[o] = lp;
          cartpole_step_one(st, i, a.seed, a.env_id0, act, r);
          a.rew[o] = r.reward;
          a.done[o] = r.done;
          if (a.ep_ret && r.done) a.ep_ret[o] = (float)r.ret;
          float* on = a.obs + ((size_t)(t + 1) * N + i) * kObs;
          reinterpret_cast<float4*>(on)[0] = make_float4(r.o_next[0], r.o_next[1], r.o_next[2], r.o_next[3]);
#pragma unroll
          for (int k = 0; k < kObs; ++k) xin[row * M::kInStride + k] = r.o_next[k];     // next policy input: the forward's LDS tile
        }
      }
      if (!tail) accumulate_ep_stats(a.ep_stats, r.done && valid, r.ret, r.len);
    }
    if (tail) break;
  }
}

}  // namespace

extern "C" {

int gymrl_rollout_cartpole(const gymrl_rollout_lunar_args* a, const gymrl_mlp_desc* policy, void* stream) {
  if (!a || !policy || !a->env_state || !a->obs || !a->act || !a->logp || !a->val || !a->rew || !a->done ||
      !a->next_value || a->n_envs <= 0 || a->T <= 0 || a->t0 < 0 || a->nsteps < 0 || a->t0 + a->nsteps > a->T)
    return -22;
  if (a->gae_running && !a->gae_workspace) return -22;
  if (reinterpret_cast<uintptr_t>(a->obs) & 15) return -22;
  // the policy must be a 2-output network on the CartPole observation: logits [2] then value [1]
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
    if (hipFuncSetAttribute((const void*)rollout_cartpole_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDynBytes) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr_set = true;
  }
  const int blocks = (a->n_envs + M::kRows - 1) / M::kRows;
  hipLaunchKernelGGL(rollout_cartpole_kernel, dim3(blocks), dim3(kThreads), kDynBytes, (hipStream_t)stream, *a, *policy);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
