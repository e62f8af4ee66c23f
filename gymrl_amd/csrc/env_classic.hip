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
#include "env_common.hpp"

using namespace gymrl;

namespace {

// ------------------------------------------------------------- CartPole ----
struct CartPoleState {
  double *x, *xd, *th, *thd;
  EpisodeFields ep;
  __host__ __device__ CartPoleState(void* buf, int n) {
    Carver c(buf, n);
    x = c.take<double>(); xd = c.take<double>(); th = c.take<double>(); thd = c.take<double>();
    ep.ep_ret = c.take<double>(); ep.ep_len = c.take<int32_t>(); ep.episode = c.take<uint32_t>();
    bytes = c.off;
  }
  size_t bytes;
};

__device__ __forceinline__ void cartpole_draw(uint64_t seed, uint64_t env, uint32_t episode,
                                              double (&s)[4]) {
  // reset: U(-0.05, 0.05)^4 in float64 (gymnasium CartPoleEnv.reset)
  const u32x4 a = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 0u);
  const u32x4 b = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 1u);
  s[0] = -0.05 + 0.1 * u01d(a.x, a.y);
  s[1] = -0.05 + 0.1 * u01d(a.z, a.w);
  s[2] = -0.05 + 0.1 * u01d(b.x, b.y);
  s[3] = -0.05 + 0.1 * u01d(b.z, b.w);
}

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
  bool done = false;
  double ret = 0.0; int len = 0;
  if (valid) {
    double x = st.x[i], xd = st.xd[i], th = st.th[i], thd = st.thd[i];
    const double force = action[i] == 1 ? 10.0 : -10.0;
    double s, c;
    det_sincos(th, &s, &c);                          // not ocml: reproducible on the host bit for bit
    const double temp = (force + 0.05 * (thd * thd) * s) / 1.1;
    const double thacc = (9.8 * s - c * temp) / (0.5 * (4.0 / 3.0 - 0.1 * (c * c) / 1.1));
    const double xacc = temp - 0.05 * thacc * c / 1.1;
    x = x + 0.02 * xd; xd = xd + 0.02 * xacc;
    th = th + 0.02 * thd; thd = thd + 0.02 * thacc;
    const double th_lim = 12.0 * 2.0 * 3.14159265358979323846 / 360.0;
    const bool terminated = x < -2.4 || x > 2.4 || th < -th_lim || th > th_lim;
    len = st.ep.ep_len[i] + 1;
    const bool truncated = len >= 500;
    done = terminated || truncated;
    ret = st.ep.ep_ret[i] + 1.0;
    rew_out[i] = 1.0f;
    terminated_out[i] = terminated; truncated_out[i] = truncated;
    if (done_out) done_out[i] = done;
    const float4 o = make_float4((float)x, (float)xd, (float)th, (float)thd);
    if (term_obs_out) reinterpret_cast<float4*>(term_obs_out)[i] = o;
    if (done) {
      if (ep_ret_out) ep_ret_out[i] = (float)ret;
      if (ep_len_out) ep_len_out[i] = len;
      const uint32_t e = st.ep.episode[i] + 1u;
      double r[4];
      cartpole_draw(seed, (uint64_t)(env_id0 + i), e, r);
      st.x[i] = r[0]; st.xd[i] = r[1]; st.th[i] = r[2]; st.thd[i] = r[3];
      st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = e;
      reinterpret_cast<float4*>(obs_out)[i] = make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    } else {
      st.x[i] = x; st.xd[i] = xd; st.th[i] = th; st.thd[i] = thd;
      st.ep.ep_ret[i] = ret; st.ep.ep_len[i] = len;
      reinterpret_cast<float4*>(obs_out)[i] = o;
    }
  }
  accumulate_ep_stats(ep_stats, done, ret, len);
}

// ------------------------------------------------------------- Pendulum ----
struct PendulumState {
  double *th, *thd;
  EpisodeFields ep;
  size_t bytes;
  __host__ __device__ PendulumState(void* buf, int n) {
    Carver c(buf, n);
    th = c.take<double>(); thd = c.take<double>();
    ep.ep_ret = c.take<double>(); ep.ep_len = c.take<int32_t>(); ep.episode = c.take<uint32_t>();
    bytes = c.off;
  }
};

__device__ __forceinline__ void pendulum_draw(uint64_t seed, uint64_t env, uint32_t episode,
                                              double& th, double& thd) {
  const u32x4 a = philox4x32(seed, (uint32_t)env, (uint32_t)(env >> 32), episode, RNG_ENV_RESET | 0u);
  const double pi = 3.14159265358979323846;
  th = -pi + (2.0 * pi) * u01d(a.x, a.y);
  thd = -1.0 + 2.0 * u01d(a.z, a.w);
}

__device__ __forceinline__ void pendulum_obs(double th, double thd, float (&o)[3]) {
  double s, c;
  det_sincos(th, &s, &c);
  o[0] = (float)c; o[1] = (float)s; o[2] = (float)thd;
}

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
  bool done = false;
  double ret = 0.0; int len = 0;
  float o_next[3] = {0.f, 0.f, 0.f}, o_term[3] = {0.f, 0.f, 0.f};
  if (valid) {
    const double pi = 3.14159265358979323846;
    double th = st.th[i], thd = st.thd[i];
    double u = (double)action[i];
    u = u < -2.0 ? -2.0 : (u > 2.0 ? 2.0 : u);
    // angle_normalize(x) = ((x + pi) mod 2pi) - pi with python's floor-mod
    double a = th + pi;
    a = a - floor(a / (2.0 * pi)) * (2.0 * pi);
    const double an = a - pi;
    const double cost = an * an + 0.1 * (thd * thd) + 0.001 * (u * u);
    double sin_th, cos_th;
    det_sincos(th, &sin_th, &cos_th);
    double nthd = thd + (15.0 * sin_th + 3.0 * u) * 0.05;    // 3g/(2l) = 15, 3/(ml^2) = 3
    nthd = nthd < -8.0 ? -8.0 : (nthd > 8.0 ? 8.0 : nthd);
    const double nth = th + nthd * 0.05;
    len = st.ep.ep_len[i] + 1;
    const bool truncated = len >= 200;
    done = truncated;
    ret = st.ep.ep_ret[i] + (-cost);
    rew_out[i] = (float)(-cost);
    terminated_out[i] = 0; truncated_out[i] = truncated;
    if (done_out) done_out[i] = done;
    pendulum_obs(nth, nthd, o_term);
    if (done) {
      if (ep_ret_out) ep_ret_out[i] = (float)ret;
      if (ep_len_out) ep_len_out[i] = len;
      const uint32_t e = st.ep.episode[i] + 1u;
      double rth, rthd;
      pendulum_draw(seed, (uint64_t)(env_id0 + i), e, rth, rthd);
      st.th[i] = rth; st.thd[i] = rthd;
      st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = e;
      pendulum_obs(rth, rthd, o_next);
    } else {
      st.th[i] = nth; st.thd[i] = nthd;
      st.ep.ep_ret[i] = ret; st.ep.ep_len[i] = len;
      o_next[0] = o_term[0]; o_next[1] = o_term[1]; o_next[2] = o_term[2];
    }
  }
  const int nv = min(kEnvBlock, n - blockIdx.x * kEnvBlock);
  const size_t base = (size_t)blockIdx.x * kEnvBlock * 3;
  store_obs_tile<3>(obs_out + base, o_next, tile, threadIdx.x, nv);
  if (term_obs_out) store_obs_tile<3>(term_obs_out + base, o_term, tile, threadIdx.x, nv);
  accumulate_ep_stats(ep_stats, done, ret, len);
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
