// replay.hip — device-resident replay ring, n-step window, uniform index draws.
//
//   D2/A3  gymrl_replay_append / _gather   dqn_cartpole.py:68-88, sac_pendulum.py:128-148,
//                                          utils/buffer.py:105-135
//   S2     gymrl_nstep_push               rainbow_dqn_cartpole.py:179-218
//          gymrl_uniform_indices          stand-in for random.sample (dqn_cartpole.py:76)
//
// The ring is SoA [cap][...] in HBM.  Appends are coalesced row writes (N rows per
// vector step); gathers read one random row per sample per field — rows are 12-32 B, so
// the gather is bound by random-access line fetches, not streaming bandwidth.
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

using namespace gymrl;

namespace {

constexpr int kBlock = 256;
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__global__ __launch_bounds__(kBlock) void replay_append_kernel(
    float* __restrict__ state, uint32_t* __restrict__ action, float* __restrict__ reward,
    float* __restrict__ next_state, uint8_t* __restrict__ flag, int64_t cap, int64_t cursor, int D,
    int AW, int n, const float* __restrict__ s_state, const uint32_t* __restrict__ s_action,
    const float* __restrict__ s_reward, const float* __restrict__ s_next,
    const uint8_t* __restrict__ s_flag, const int64_t* __restrict__ cursor_dev) {
  if (cursor_dev) cursor = cursor_dev[0];      // recorded into a hipGraph: the cursor of this replay lives on the device
  // one thread per (row, word): words = D state + D next + AW action + reward + flag
  const int W = 2 * D + AW + 2;
  const int64_t total = (int64_t)n * W;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int i = (int)(t / W), k = (int)(t % W);
    const int64_t row = (cursor + i) % cap;
    if (k < D) state[row * D + k] = s_state[(int64_t)i * D + k];
    else if (k < 2 * D) next_state[row * D + (k - D)] = s_next[(int64_t)i * D + (k - D)];
    else if (k < 2 * D + AW) action[row * AW + (k - 2 * D)] = s_action[(int64_t)i * AW + (k - 2 * D)];
    else if (k == 2 * D + AW) reward[row] = s_reward[i];
    else flag[row] = s_flag[i];
  }
}

__global__ __launch_bounds__(kBlock) void replay_gather_kernel(
    const float* __restrict__ state, const uint32_t* __restrict__ action,
    const float* __restrict__ reward, const float* __restrict__ next_state,
    const uint8_t* __restrict__ flag, const int32_t* __restrict__ idx, int B, int D, int AW,
    float* __restrict__ o_state, uint32_t* __restrict__ o_action, float* __restrict__ o_reward,
    float* __restrict__ o_next, float* __restrict__ o_flag) {
  const int W = 2 * D + AW + 2;
  const int64_t total = (int64_t)B * W;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int b = (int)(t / W), k = (int)(t % W);
    const int64_t row = idx[b];
    if (k < D) o_state[(int64_t)b * D + k] = state[row * D + k];
    else if (k < 2 * D) o_next[(int64_t)b * D + (k - D)] = next_state[row * D + (k - D)];
    else if (k < 2 * D + AW) o_action[(int64_t)b * AW + (k - 2 * D)] = action[row * AW + (k - 2 * D)];
    else if (k == 2 * D + AW) o_reward[b] = reward[row];
    else o_flag[b] = (float)flag[row];           // dones/terminal become float32 (dqn_cartpole.py:155)
  }
}

// random.sample(buffer, B) draws B DISTINCT rows: idx[b] = the b-th element of a keyed permutation of [0, size)
// (gymrl_device.hpp keyed_permute, tag RNG_REPLAY folded into the key) — a uniform sample without replacement
// with no rejection loop and no bookkeeping between lanes.
struct UniformDev { uint64_t counter; int64_t size; };
__global__ __launch_bounds__(kBlock) void uniform_indices_kernel(uint64_t seed, uint64_t counter,
                                                                 uint32_t size, int B, int a, int bbits,
                                                                 int32_t* __restrict__ idx,
                                                                 const UniformDev* __restrict__ dev) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  if (dev) {                                   // per-replay (counter, size) from the device; same split of the bits as the host's
    counter = dev->counter;
    size = (uint32_t)dev->size;
    int bits = 2;
    while (((int64_t)1 << bits) < (int64_t)size) ++bits;
    a = bits / 2; bbits = bits - bits / 2;
  }
  idx[b] = (int32_t)keyed_permute((uint32_t)b, size, a, bbits, seed ^ 0x5265706C61794944ull, counter);
}

// One lane = one env.  Window slot of entry "i-th oldest" once full: (slot + 1 + i) % n.
__global__ __launch_bounds__(kBlock) void nstep_push_kernel(
    float* __restrict__ w_state, int32_t* __restrict__ w_action, float* __restrict__ w_reward,
    float* __restrict__ w_next, uint8_t* __restrict__ w_terminal, uint8_t* __restrict__ w_done,
    int n_steps, int slot, int emit, int N, int D, double gamma, const float* __restrict__ obs,
    const int32_t* __restrict__ action, const float* __restrict__ reward,
    const float* __restrict__ next_obs, const uint8_t* __restrict__ terminal,
    const uint8_t* __restrict__ done, float* __restrict__ r_state, uint32_t* __restrict__ r_action,
    float* __restrict__ r_reward, float* __restrict__ r_next, uint8_t* __restrict__ r_flag,
    int64_t cap, int64_t cursor, const int64_t* __restrict__ dev, const int32_t* __restrict__ ep_len,
    int max_episode_steps) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= N) return;
  if (dev) {                                   // {pushes, cursor} of this replay
    const int64_t pushes = dev[0];
    slot = (int)(pushes % n_steps);
    emit = pushes + 1 >= n_steps ? 1 : 0;
    cursor = dev[1];
  }
  // deque.append(transition) — :186-187
  const size_t so = ((size_t)slot * N + e);
  for (int k = 0; k < D; ++k) {
    w_state[so * D + k] = obs[(size_t)e * D + k];
    w_next[so * D + k] = next_obs[(size_t)e * D + k];
  }
  w_action[so] = action[e]; w_reward[so] = reward[e];
  // terminal NULL: rainbow_dqn_cartpole.py:376 — done and not the last step of the time limit, by the step INDEX
  w_terminal[so] = terminal ? terminal[e] : (uint8_t)((done[e] != 0 && ep_len[e] != max_episode_steps) ? 1 : 0);
  w_done[so] = done[e];
  if (!emit) return;
  // _get_n_step_transition — :207-218
  const int oldest = (slot + 1) % n_steps;
  int src = slot;                              // (next_state, terminal) default: newest entry
  double R = 0.0;
  for (int i = n_steps - 1; i >= 0; --i) {
    const int s = (oldest + i) % n_steps;
    const size_t o = (size_t)s * N + e;
    const double d = w_done[o] ? 1.0 : 0.0;
    R = (double)w_reward[o] + gamma * (1.0 - d) * R;
    if (w_done[o]) src = s;                    // ends on the EARLIEST done in the window
  }
  const int64_t row = (cursor + e) % cap;
  const size_t oo = (size_t)oldest * N + e, ss = (size_t)src * N + e;
  for (int k = 0; k < D; ++k) {
    r_state[row * D + k] = w_state[oo * D + k];
    r_next[row * D + k] = w_next[ss * D + k];
  }
  r_action[row] = (uint32_t)w_action[oo];
  r_reward[row] = (float)R;
  r_flag[row] = w_terminal[ss];
}

}  // namespace

extern "C" {

int gymrl_replay_append(float* state, uint32_t* action, float* reward, float* next_state,
                        uint8_t* flag, int64_t cap, int64_t cursor, int D, int AW, int n,
                        const float* src_state, const void* src_action, const float* src_reward,
                        const float* src_next_state, const uint8_t* src_flag, const int64_t* cursor_dev,
                        void* stream_) {
  if (!state || !action || !reward || !next_state || !flag || !src_state || !src_action ||
      !src_reward || !src_next_state || !src_flag || cap <= 0 || cursor < 0 || D <= 0 || AW <= 0 ||
      n < 0 || n > cap)
    return -22;
  if (n == 0) return 0;
  const int64_t total = (int64_t)n * (2 * D + AW + 2);
  int nb = cdiv(total, kBlock);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(replay_append_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream_, state, action,
                     reward, next_state, flag, cap, cursor, D, AW, n, src_state,
                     (const uint32_t*)src_action, src_reward, src_next_state, src_flag, cursor_dev);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_replay_gather(const float* state, const uint32_t* action, const float* reward,
                        const float* next_state, const uint8_t* flag, const int32_t* idx, int B,
                        int D, int AW, float* state_out, void* action_out, float* reward_out,
                        float* next_state_out, float* flag_out, void* stream_) {
  if (!state || !action || !reward || !next_state || !flag || !idx || !state_out || !action_out ||
      !reward_out || !next_state_out || !flag_out || B < 0 || D <= 0 || AW <= 0)
    return -22;
  if (B == 0) return 0;
  const int64_t total = (int64_t)B * (2 * D + AW + 2);
  int nb = cdiv(total, kBlock);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(replay_gather_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream_, state, action,
                     reward, next_state, flag, idx, B, D, AW, state_out, (uint32_t*)action_out,
                     reward_out, next_state_out, flag_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_uniform_indices(uint64_t seed, uint64_t counter, int64_t size, int B, int32_t* idx_out,
                          const void* dev, void* stream_) {
  if (!idx_out || size <= 0 || size > 0x7FFFFFFF || B < 0 || B > size) return -22;
  if (B == 0) return 0;
  int bits = 2;
  while (((int64_t)1 << bits) < size) ++bits;
  hipLaunchKernelGGL(uniform_indices_kernel, dim3(cdiv(B, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream_, seed, counter, (uint32_t)size, B, bits / 2, bits - bits / 2, idx_out,
                     static_cast<const UniformDev*>(dev));
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_nstep_push(float* w_state, int32_t* w_action, float* w_reward, float* w_next,
                     uint8_t* w_terminal, uint8_t* w_done, int n_steps, int64_t pushes, int N, int D,
                     double gamma, const float* obs, const int32_t* action, const float* reward,
                     const float* next_obs, const uint8_t* terminal, const uint8_t* done,
                     float* r_state, uint32_t* r_action, float* r_reward, float* r_next,
                     uint8_t* r_flag, int64_t cap, int64_t cursor, const int64_t* dev, const int32_t* ep_len,
                     int max_episode_steps, void* stream_) {
  if (!w_state || !w_action || !w_reward || !w_next || !w_terminal || !w_done || !obs || !action ||
      !reward || !next_obs || (!terminal && !ep_len) || !done || !r_state || !r_action || !r_reward || !r_next ||
      !r_flag || n_steps <= 0 || pushes < 0 || N < 0 || D <= 0 || cap < N || cursor < 0)
    return -22;
  if (N == 0) return 0;
  const int slot = (int)(pushes % n_steps);
  const int emit = (pushes + 1 >= n_steps) ? 1 : 0;
  hipLaunchKernelGGL(nstep_push_kernel, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_,
                     w_state, w_action, w_reward, w_next, w_terminal, w_done, n_steps, slot, emit, N, D,
                     gamma, obs, action, reward, next_obs, terminal, done, r_state, r_action, r_reward,
                     r_next, r_flag, cap, cursor, dev, ep_len, max_episode_steps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -1000 - (int)e;
  return emit;
}

}  // extern "C"
