// env_classic.hip — CartPole-v1 and Pendulum-v1 batched steppers + the
// gymrl_env_* C-ABI dispatch (LunarLander lives in env_lunar.hip).
//
// Replaces gym.make(...).reset()/.step() as called from dqn_cartpole.py:176,181,
// rainbow_dqn_cartpole.py:369,373, sac_pendulum.py:275,281, utils/runner.py:111,123.
// The env arithmetic itself is gymnasium's (third-party, not under the reference
// tree, version unpinned: SURVEY.md §8c.2) — dynamics follow its published
// classic_control definitions; state is float64 like gymnasium's, observations
// are the float32 cast.
//
// One lane = one env; SoA state [field][N]; per-lane work is ~40 flop, so the
// kernels are trivially HBM/launch bound (58 / 38 algorithmic bytes per env-step).
#include "env_classic_device.hpp"

using namespace gymrl;

namespace {

// ------------------------------------------------------------- CartPole ----
// (state layout, reset draws and the per-env step: env_classic_device.hpp)
__global__ __launch_bounds__(kEnvBlock) void cartpole_reset_kernel(void* buf, int n, uint64_t seed,
                                                                   int64_t env_id0,
                                                                   float* __restrict__ obs_out) {
  CartPoleState st(buf, n);
  const int i = blockIdx.x * kEnvBlock + threadIdx.x;
  if (i >= n) return;
  double s[4];
  cartpole_draw(seed, (uint64_t)(env_id0 + i), 0u, s);
  st.x[i] = s[0]; st.xd[i] = s[1]; st.th[i] = s[2]; st.thd[i] = s[3];
  st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = 0u;
  reinterpret_cast<float4*>(obs_out)[i] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
}

__global__ __launch_bounds__(kEnvBlock) void cartpole_step_kernel(
    void* buf, int n, uint64_t seed, int64_t env_id0, const int32_t* __restrict__ action,
    float* __restrict__ obs_out, float* __restrict__ term_obs_out, float* __restrict__ rew_out,
    uint8_t* __restrict__ terminated_out, uint8_t* __restrict__ truncated_out,
    uint8_t* __restrict__ done_out, float* __restrict__ ep_ret_out,
    int32_t* __restrict__ ep_len_out, double* __restrict__ ep_stats) {
  CartPoleState st(buf, n);
  const int i = blockIdx.x * kEnvBlock + threadIdx.x;
  const bool valid = i < n;
  ClassicStep<4> r;
  r.done = false; r.ret = 0.0; r.len = 0;
  if (valid) {
    cartpole_step_one(st, i, seed, env_id0, action[i], r);
    rew_out[i] = r.reward;
    terminated_out[i] = r.terminated; truncated_out[i] = r.truncated;
    if (done_out) done_out[i] = r.done;
    if (term_obs_out) reinterpret_cast<float4*>(term_obs_out)[i] = make_float4(r.o_term[0], r.o_term[1], r.o_term[2], r.o_term[3]);
    if (r.done) {
      if (ep_ret_out) ep_ret_out[i] = (float)r.ret;
      if (ep_len_out) ep_len_out[i] = r.len;
    }
    reinterpret_cast<float4*>(obs_out)[i] = make_float4(r.o_next[0], r.o_next[1], r.o_next[2], r.o_next[3]);
  }
  accumulate_ep_stats(ep_stats, r.done, r.ret, r.len);
}

// ------------------------------------------------------------- Pendulum ----
__global__ __launch_bounds__(kEnvBlock) void pendulum_reset_kernel(void* buf, int n, uint64_t seed,
                                                                   int64_t env_id0,
                                                                   float* __restrict__ obs_out) {
  __shared__ __attribute__((aligned(16))) float tile[kEnvBlock * 3];
  PendulumState st(buf, n);
  const int i = blockIdx.x * kEnvBlock + threadIdx.x;
  float o[3] = {0.f, 0.f, 0.f};
  if (i < n) {
    double th, thd;
    pendulum_draw(seed, (uint64_t)(env_id0 + i), 0u, th, thd);
    st.th[i] = th; st.thd[i] = thd;
    st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = 0u;
    pendulum_obs(th, thd, o);
  }
  const int nv = min(kEnvBlock, n - blockIdx.x * kEnvBlock);
  store_obs_tile<3>(obs_out + (size_t)blockIdx.x * kEnvBlock * 3, o, tile, threadIdx.x, nv);
}

__global__ __launch_bounds__(kEnvBlock) void pendulum_step_kernel(
    void* buf, int n, uint64_t seed, int64_t env_id0, const float* __restrict__ action,
    float* __restrict__ obs_out, float* __restrict__ term_obs_out, float* __restrict__ rew_out,
    uint8_t* __restrict__ terminated_out, uint8_t* __restrict__ truncated_out,
    uint8_t* __restrict__ done_out, float* __restrict__ ep_ret_out,
    int32_t* __restrict__ ep_len_out, double* __restrict__ ep_stats) {
  __shared__ __attribute__((aligned(16))) float tile[kEnvBlock * 3];
  PendulumState st(buf, n);
  const int i = blockIdx.x * kEnvBlock + threadIdx.x;
  const bool valid = i < n;
  ClassicStep<3> r;
  r.done = false; r.ret = 0.0; r.len = 0;
  r.o_next[0] = r.o_next[1] = r.o_next[2] = 0.f; r.o_term[0] = r.o_term[1] = r.o_term[2] = 0.f;
  if (valid) {
    pendulum_step_one(st, i, seed, env_id0, action[i], r);
    rew_out[i] = r.reward;
    terminated_out[i] = 0; truncated_out[i] = r.truncated;
    if (done_out) done_out[i] = r.done;
    if (r.done) {
      if (ep_ret_out) ep_ret_out[i] = (float)r.ret;
      if (ep_len_out) ep_len_out[i] = r.len;
    }
  }
  const int nv = min(kEnvBlock, n - blockIdx.x * kEnvBlock);
  const size_t base = (size_t)blockIdx.x * kEnvBlock * 3;
  store_obs_tile<3>(obs_out + base, r.o_next, tile, threadIdx.x, nv);
  if (term_obs_out) store_obs_tile<3>(term_obs_out + base, r.o_term, tile, threadIdx.x, nv);
  accumulate_ep_stats(ep_stats, r.done, r.ret, r.len);
}

// Episodes a trainer abandons at its own step cap (dqn_cartpole.py:178 `for step in range(cfg.max_steps)` below the
// env's TimeLimit): an env whose running episode has reached `cap` steps starts its next episode — no done flag, the
// transition just stored keeps the real next observation — and reports the abandoned episode's return / length.
__global__ __launch_bounds__(kEnvBlock) void classic_abandon_kernel(int kind, void* buf, int n, uint64_t seed, int64_t env_id0,
                                                                    int cap, float* __restrict__ obs, uint8_t* __restrict__ flag_out,
                                                                    float* __restrict__ ep_ret_out, int32_t* __restrict__ ep_len_out,
                                                                    double* __restrict__ ep_stats) {
  const int i = blockIdx.x * kEnvBlock + threadIdx.x;
  bool hit = false;
  double ret = 0.0; int len = 0;
  if (i < n) {
    EpisodeFields ep = kind == GYMRL_ENV_CARTPOLE ? CartPoleState(buf, n).ep : PendulumState(buf, n).ep;
    len = ep.ep_len[i];
    hit = len >= cap;
    if (hit) {
      ret = ep.ep_ret[i];
      const uint32_t e = ep.episode[i] + 1u;
      if (kind == GYMRL_ENV_CARTPOLE) {
        CartPoleState st(buf, n);
        double r[4];
        cartpole_draw(seed, (uint64_t)(env_id0 + i), e, r);
        st.x[i] = r[0]; st.xd[i] = r[1]; st.th[i] = r[2]; st.thd[i] = r[3];
        reinterpret_cast<float4*>(obs)[i] = make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
      } else {
        PendulumState st(buf, n);
        double th, thd;
        pendulum_draw(seed, (uint64_t)(env_id0 + i), e, th, thd);
        st.th[i] = th; st.thd[i] = thd;
        float o[3];
        pendulum_obs(th, thd, o);
        obs[3 * (size_t)i] = o[0]; obs[3 * (size_t)i + 1] = o[1]; obs[3 * (size_t)i + 2] = o[2];
      }
      ep.ep_ret[i] = 0.0; ep.ep_len[i] = 0; ep.episode[i] = e;
      if (ep_ret_out) ep_ret_out[i] = (float)ret;
      if (ep_len_out) ep_len_out[i] = len;
    }
    if (flag_out) flag_out[i] = flag_out[i] | (uint8_t)hit;
  }
  accumulate_ep_stats(ep_stats, hit, ret, len);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

extern "C" {

int gymrl_abi_version(void) { return GYMRL_ABI_VERSION; }

int gymrl_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return __builtin_strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int gymrl_env_obs_dim(int kind) {
  return kind == GYMRL_ENV_CARTPOLE ? 4 : kind == GYMRL_ENV_PENDULUM ? 3 : kind == GYMRL_ENV_LUNARLANDER ? 8 : -22;
}
int gymrl_env_act_dim(int kind) {
  return kind == GYMRL_ENV_CARTPOLE ? 2 : kind == GYMRL_ENV_PENDULUM ? 1 : kind == GYMRL_ENV_LUNARLANDER ? 4 : -22;
}
int gymrl_env_is_discrete(int kind) {
  return kind == GYMRL_ENV_PENDULUM ? 0 : (kind == GYMRL_ENV_CARTPOLE || kind == GYMRL_ENV_LUNARLANDER) ? 1 : -22;
}
int gymrl_env_max_steps(int kind) {
  return kind == GYMRL_ENV_CARTPOLE ? 500 : kind == GYMRL_ENV_PENDULUM ? 200 : kind == GYMRL_ENV_LUNARLANDER ? 1000 : -22;
}

size_t gymrl_env_state_bytes(int kind, int n_envs) {
  if (n_envs <= 0) return 0;
  switch (kind) {
    case GYMRL_ENV_CARTPOLE: return CartPoleState(nullptr, n_envs).bytes;
    case GYMRL_ENV_PENDULUM: return PendulumState(nullptr, n_envs).bytes;
    case GYMRL_ENV_LUNARLANDER: return lunar_state_bytes(n_envs);
    default: return 0;
  }
}

int gymrl_env_reset(int kind, void* state, int n, uint64_t seed, int64_t env_id0, float* obs_out,
                    void* stream_) {
  if (!state || !obs_out || n < 0 || !aligned(state, 256) || !aligned(obs_out, 16)) return -22;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream_;
  switch (kind) {
    case GYMRL_ENV_CARTPOLE:
      hipLaunchKernelGGL(cartpole_reset_kernel, dim3(cdiv(n, kEnvBlock)), dim3(kEnvBlock), 0, s,
                         state, n, seed, env_id0, obs_out);
      break;
    case GYMRL_ENV_PENDULUM:
      hipLaunchKernelGGL(pendulum_reset_kernel, dim3(cdiv(n, kEnvBlock)), dim3(kEnvBlock), 0, s,
                         state, n, seed, env_id0, obs_out);
      break;
    case GYMRL_ENV_LUNARLANDER:
      return lunar_reset(state, n, seed, env_id0, obs_out, s);
    default: return -22;
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_env_abandon(int kind, void* state, int n, uint64_t seed, int64_t env_id0, int cap, float* obs_inout,
                      uint8_t* flag_inout, float* ep_ret_out, int32_t* ep_len_out, double* ep_stats, void* stream_) {
  if (!state || !obs_inout || n < 0 || cap <= 0 || !aligned(state, 256) || !aligned(obs_inout, 16)) return -22;
  if (kind != GYMRL_ENV_CARTPOLE && kind != GYMRL_ENV_PENDULUM) return -22;    // no reference off-policy script runs LunarLander
  if (n == 0) return 0;
  hipLaunchKernelGGL(classic_abandon_kernel, dim3(cdiv(n, kEnvBlock)), dim3(kEnvBlock), 0, (hipStream_t)stream_, kind, state,
                     n, seed, env_id0, cap, obs_inout, flag_inout, ep_ret_out, ep_len_out, ep_stats);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_env_refill(int kind, void* state, int n, uint64_t seed, int64_t env_id0, void* stream_) {
  if (!state || n < 0 || !aligned(state, 256)) return -22;
  if (n == 0) return 0;
  if (kind == GYMRL_ENV_LUNARLANDER) return lunar_refill(state, n, seed, env_id0, (hipStream_t)stream_);
  return (kind == GYMRL_ENV_CARTPOLE || kind == GYMRL_ENV_PENDULUM) ? 0 : -22;   // cheap resets: nothing to prepare
}

int gymrl_env_step(int kind, void* state, int n, uint64_t seed, int64_t env_id0, const void* action,
                   float* obs_out, float* term_obs_out, float* rew_out, uint8_t* terminated_out,
                   uint8_t* truncated_out, uint8_t* done_out, float* ep_ret_out,
                   int32_t* ep_len_out, double* ep_stats, void* stream_) {
  if (!state || !action || !obs_out || !rew_out || !terminated_out || !truncated_out || n < 0)
    return -22;
  if (!aligned(state, 256) || !aligned(obs_out, 16) || (term_obs_out && !aligned(term_obs_out, 16)))
    return -22;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream_;
  switch (kind) {
    case GYMRL_ENV_CARTPOLE:
      hipLaunchKernelGGL(cartpole_step_kernel, dim3(cdiv(n, kEnvBlock)), dim3(kEnvBlock), 0, s,
                         state, n, seed, env_id0, (const int32_t*)action, obs_out, term_obs_out,
                         rew_out, terminated_out, truncated_out, done_out, ep_ret_out, ep_len_out,
                         ep_stats);
      break;
    case GYMRL_ENV_PENDULUM:
      hipLaunchKernelGGL(pendulum_step_kernel, dim3(cdiv(n, kEnvBlock)), dim3(kEnvBlock), 0, s,
                         state, n, seed, env_id0, (const float*)action, obs_out, term_obs_out,
                         rew_out, terminated_out, truncated_out, done_out, ep_ret_out, ep_len_out,
                         ep_stats);
      break;
    case GYMRL_ENV_LUNARLANDER:
      return lunar_step(state, n, seed, env_id0, (const int32_t*)action, obs_out, term_obs_out,
                        rew_out, terminated_out, truncated_out, done_out, ep_ret_out, ep_len_out,
                        ep_stats, s);
    default: return -22;
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
