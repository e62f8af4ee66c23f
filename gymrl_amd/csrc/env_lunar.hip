// env_lunar.hip — LunarLander-v3 batched stepper: kernels and launchers (the solver itself is in
// env_lunar_device.hpp, which carries the design notes).
#include "env_lunar_device.hpp"

using namespace gymrl;
using namespace gymrl::lunar;

namespace {

__global__ __launch_bounds__(kEnvBlock) void lunar_reset_kernel(void* buf, int n, uint64_t seed,
                                                                int64_t env_id0,
                                                                float* __restrict__ obs_out) {
  __shared__ uint32_t lds_words[kLdsWords * kEnvBlock];
  const Lds lds{lds_words + threadIdx.x};
  LunarState st(buf, n);
  const int role = threadIdx.x & 3;
  const int i = blockIdx.x * kEnvsPerBlock + (threadIdx.x >> 2);
  if (i >= n) return;                       // whole quads leave together
  World W;
  float fx, fy, rew, o[8]; bool term;
  init_episode(W, lds, seed, (uint64_t)(env_id0 + i), 0u, fx, fy);
  env_step_once(W, lds, role, 0, seed, (uint64_t)(env_id0 + i), 0u, 0u, fx, fy, o, rew, term);   // reset() ends with step(0)
  world_io(W, lds, role, st.words, n, i, true);
  if (role == 0) {
    st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = 0u;
    st.spare_episode[i] = kNoSpare;
  }
  store_obs_quad(obs_out, i, role, o);
}

// Builds the spare world of episode (current + 1) for every env that lacks one.
__global__ __launch_bounds__(kEnvBlock) void lunar_refill_kernel(void* buf, int n, uint64_t seed,
                                                                 int64_t env_id0) {
  __shared__ uint32_t lds_words[kLdsWords * kEnvBlock];
  const Lds lds{lds_words + threadIdx.x};
  LunarState st(buf, n);
  const int role = threadIdx.x & 3;
  const int i = blockIdx.x * kEnvsPerBlock + (threadIdx.x >> 2);
  if (i >= n) return;
  const uint32_t want = st.ep.episode[i] + 1u;
  lunar_refill_quad(st, lds, n, i, role, st.spare_episode[i] != want, want, seed, env_id0);
}

__global__ __launch_bounds__(kEnvBlock) void lunar_step_kernel(
    void* buf, int n, uint64_t seed, int64_t env_id0, const int32_t* __restrict__ action,
    float* __restrict__ obs_out, float* __restrict__ term_obs_out, float* __restrict__ rew_out,
    uint8_t* __restrict__ terminated_out, uint8_t* __restrict__ truncated_out,
    uint8_t* __restrict__ done_out, float* __restrict__ ep_ret_out,
    int32_t* __restrict__ ep_len_out, double* __restrict__ ep_stats) {
  __shared__ uint32_t lds_words[kLdsWords * kEnvBlock];
  const Lds lds{lds_words + threadIdx.x};
  const LunarState st(buf, n);
  const int role = threadIdx.x & 3;
  const int i = blockIdx.x * kEnvsPerBlock + (threadIdx.x >> 2);
  const bool valid = i < n;
  const StepOut out{obs_out, term_obs_out, rew_out, terminated_out, truncated_out, done_out, ep_ret_out, ep_len_out, ep_stats};
  float o_next[8];
  lunar_step_quad(st, lds, n, i, role, valid, valid ? action[i] : 0, seed, env_id0, out, o_next);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

namespace gymrl {

size_t lunar_state_bytes(int n) { return LunarState(nullptr, n).bytes; }

int lunar_reset(void* state, int n, uint64_t seed, int64_t env_id0, float* obs_out, hipStream_t s) {
  hipLaunchKernelGGL(lunar_reset_kernel, dim3(cdiv(n, kEnvsPerBlock)), dim3(kEnvBlock), 0, s, state, n, seed,
                     env_id0, obs_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int lunar_refill(void* state, int n, uint64_t seed, int64_t env_id0, hipStream_t s) {
  hipLaunchKernelGGL(lunar_refill_kernel, dim3(cdiv(n, kEnvsPerBlock)), dim3(kEnvBlock), 0, s, state, n, seed, env_id0);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int lunar_step(void* state, int n, uint64_t seed, int64_t env_id0, const int32_t* action,
               float* obs_out, float* term_obs_out, float* rew_out, uint8_t* terminated_out,
               uint8_t* truncated_out, uint8_t* done_out, float* ep_ret_out, int32_t* ep_len_out,
               double* ep_stats, hipStream_t s) {
  hipLaunchKernelGGL(lunar_step_kernel, dim3(cdiv(n, kEnvsPerBlock)), dim3(kEnvBlock), 0, s, state, n, seed,
                     env_id0, action, obs_out, term_obs_out, rew_out, terminated_out, truncated_out,
                     done_out, ep_ret_out, ep_len_out, ep_stats);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // namespace gymrl
