// env_common.hpp — shared pieces of the batched env steppers (one env per lane).
#pragma once
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

namespace gymrl {

constexpr int kEnvBlock = 64;  // one wavefront per workgroup: spreads N/64 waves over N/64 CUs

// SoA carving of the caller-owned state buffer: every field is an [N] array
// whose base is 256-B aligned, so lane i of a wave touches word i of a line.
struct Carver {
  char* base;
  size_t off;
  int n;
  __host__ __device__ Carver(void* b, int n_) : base((char*)b), off(0), n(n_) {}
  template <typename T>
  __host__ __device__ T* take(int count_per_env = 1) {
    T* p = reinterpret_cast<T*>(base + off);
    size_t bytes = sizeof(T) * (size_t)n * (size_t)count_per_env;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  }
};

// Per-env episode bookkeeping common to all kinds.
struct EpisodeFields {
  double* ep_ret;     // running undiscounted return (python float in the reference loop)
  int32_t* ep_len;    // steps taken in the current episode (TimeLimit counter)
  uint32_t* episode;  // episode index: Philox counter word for resets
};

// Stage a wave's [64][D] observation tile through LDS so the global stores are
// whole contiguous 16-B-per-lane rows (1 KiB per store instruction) instead of
// D strided dword stores.  `tile` = kEnvBlock*D floats of LDS.  n_valid lanes.
template <int D>
__device__ __forceinline__ void store_obs_tile(float* __restrict__ dst_base, const float (&o)[D],
                                               float* tile, int lane, int n_valid) {
  static_assert((kEnvBlock * D) % 4 == 0, "tile must be float4-divisible");
  if constexpr (D == 4) {
    if (lane < n_valid)
      reinterpret_cast<float4*>(dst_base)[lane] = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) tile[lane * D + k] = o[k];
    __syncthreads();
    const int nflt = n_valid * D;
    constexpr int kVec = kEnvBlock * D / 4;
#pragma unroll
    for (int j = lane; j < kVec; j += kEnvBlock) {
      if (4 * j + 3 < nflt) {
        reinterpret_cast<float4*>(dst_base)[j] = reinterpret_cast<const float4*>(tile)[j];
      } else {
        for (int e = 4 * j; e < nflt; ++e) dst_base[e] = tile[e];
      }
    }
    __syncthreads();
  }
}

// Wave-level episode statistics: ballot the done mask, one f64 atomic per field
// per wave (ppo_lunarlander.py:220-221 appends episode_reward on done).
__device__ __forceinline__ void accumulate_ep_stats(double* ep_stats, bool done, double ep_ret,
                                                    int ep_len) {
  if (!ep_stats) return;
  const unsigned long long mask = __ballot(done);
  if (mask == 0ull) return;
  double r = done ? ep_ret : 0.0;
  double l = done ? (double)ep_len : 0.0;
  r = wave_sum(r); l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(ep_stats + 0, (double)__popcll(mask));
    atomicAdd(ep_stats + 1, r);
    atomicAdd(ep_stats + 2, l);
  }
}

// launchers implemented per env kind
size_t lunar_state_bytes(int n);
int lunar_reset(void* state, int n, uint64_t seed, int64_t env_id0, float* obs_out, hipStream_t s);
int lunar_refill(void* state, int n, uint64_t seed, int64_t env_id0, hipStream_t s);
int lunar_step(void* state, int n, uint64_t seed, int64_t env_id0, const int32_t* action,
               float* obs_out, float* term_obs_out, float* rew_out, uint8_t* terminated_out,
               uint8_t* truncated_out, uint8_t* done_out, float* ep_ret_out, int32_t* ep_len_out,
               double* ep_stats, hipStream_t s);

}  // namespace gymrl
