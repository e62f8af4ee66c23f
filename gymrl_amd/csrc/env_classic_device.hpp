// env_classic_device.hpp — CartPole-v1 / Pendulum-v1: state layout, reset draws and ONE env's step as device functions.
//
// The arithmetic of gym.make("CartPole-v1" | "Pendulum-v1").step()/reset() (gymnasium classic_control, third-party:
// SURVEY.md 8c.2) for one env = one lane.  Shared by the stand-alone steppers (env_classic.hip: gymrl_env_step) and the
// fused acting kernels (offpolicy_step.hip: forward + draw + env step + ring append in one launch), so both produce the
// same bits.  State is float64 like gymnasium's, observations are the float32 cast; sin / cos are det_sincos.
#pragma once
#include "env_common.hpp"

namespace gymrl {

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

// What one env's step hands back: the observation the policy sees next (post-reset where the episode ended), the
// terminal observation (what off-policy buffers store as next_state), reward, flags, and the finished episode's
// return / length (valid where done).
template <int D>
struct ClassicStep {
  float o_next[D], o_term[D];
  float reward;
  bool terminated, truncated, done;
  double ret;
  int len;
};

// CartPole-v1: explicit Euler with the pre-step velocities, terminated beyond |x| > 2.4 or |theta| > 12 degrees, reward 1
// on every step including the terminating one, TimeLimit 500; auto-reset into episode + 1's draw.
__device__ __forceinline__ void cartpole_step_one(const CartPoleState& st, int i, uint64_t seed, int64_t env_id0, int action,
                                                  ClassicStep<4>& r) {
  double x = st.x[i], xd = st.xd[i], th = st.th[i], thd = st.thd[i];
  const double force = action == 1 ? 10.0 : -10.0;
  double s, c;
  det_sincos(th, &s, &c);                          // not ocml: reproducible on the host bit for bit
  const double temp = (force + 0.05 * (thd * thd) * s) / 1.1;
  const double thacc = (9.8 * s - c * temp) / (0.5 * (4.0 / 3.0 - 0.1 * (c * c) / 1.1));
  const double xacc = temp - 0.05 * thacc * c / 1.1;
  x = x + 0.02 * xd; xd = xd + 0.02 * xacc;
  th = th + 0.02 * thd; thd = thd + 0.02 * thacc;
  const double th_lim = 12.0 * 2.0 * 3.14159265358979323846 / 360.0;
  r.terminated = x < -2.4 || x > 2.4 || th < -th_lim || th > th_lim;
  r.len = st.ep.ep_len[i] + 1;
  r.truncated = r.len >= 500;
  r.done = r.terminated || r.truncated;
  r.ret = st.ep.ep_ret[i] + 1.0;
  r.reward = 1.0f;
  r.o_term[0] = (float)x; r.o_term[1] = (float)xd; r.o_term[2] = (float)th; r.o_term[3] = (float)thd;
  if (r.done) {
    const uint32_t e = st.ep.episode[i] + 1u;
    double d[4];
    cartpole_draw(seed, (uint64_t)(env_id0 + i), e, d);
    st.x[i] = d[0]; st.xd[i] = d[1]; st.th[i] = d[2]; st.thd[i] = d[3];
    st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = e;
    r.o_next[0] = (float)d[0]; r.o_next[1] = (float)d[1]; r.o_next[2] = (float)d[2]; r.o_next[3] = (float)d[3];
  } else {
    st.x[i] = x; st.xd[i] = xd; st.th[i] = th; st.thd[i] = thd;
    st.ep.ep_ret[i] = r.ret; st.ep.ep_len[i] = r.len;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.o_next[k] = r.o_term[k];
  }
}

// Pendulum-v1: torque clipped to +-2, cost from the PRE-step state, speed clipped to +-8, never terminates, TimeLimit 200.
__device__ __forceinline__ void pendulum_step_one(const PendulumState& st, int i, uint64_t seed, int64_t env_id0, float action,
                                                  ClassicStep<3>& r) {
  const double pi = 3.14159265358979323846;
  double th = st.th[i], thd = st.thd[i];
  double u = (double)action;
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
  r.len = st.ep.ep_len[i] + 1;
  r.terminated = false;
  r.truncated = r.len >= 200;
  r.done = r.truncated;
  r.ret = st.ep.ep_ret[i] + (-cost);
  r.reward = (float)(-cost);
  pendulum_obs(nth, nthd, r.o_term);
  if (r.done) {
    const uint32_t e = st.ep.episode[i] + 1u;
    double rth, rthd;
    pendulum_draw(seed, (uint64_t)(env_id0 + i), e, rth, rthd);
    st.th[i] = rth; st.thd[i] = rthd;
    st.ep.ep_ret[i] = 0.0; st.ep.ep_len[i] = 0; st.ep.episode[i] = e;
    pendulum_obs(rth, rthd, r.o_next);
  } else {
    st.th[i] = nth; st.thd[i] = nthd;
    st.ep.ep_ret[i] = r.ret; st.ep.ep_len[i] = r.len;
    r.o_next[0] = r.o_term[0]; r.o_next[1] = r.o_term[1]; r.o_next[2] = r.o_term[2];
  }
}

}  // namespace gymrl
