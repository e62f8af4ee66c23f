// gae.hip — Generalised Advantage Estimation kernels for gfx950.
//
//   G1  gymrl_gae            ppo_lunarlander.py:179-196   (f64 recursion)
//   G2  gymrl_gae_dw         utils/buffer.py:21-35        (f32 recursion, dw/done)
//   G3  gymrl_gae_decoupled  ppo_full_lunarlander.py:507-535
//   P5  gymrl_moments / gymrl_normalize   ppo_lunarlander.py:236, utils/buffer.py:33
//
// Layout: every array is time-major [T][N]; a wavefront's lanes are adjacent
// env instances, so each row access is one contiguous segment.
//
// HBM-bound: 17 algorithmic bytes per (t, env) element (4 rew + 4 val + 1 done
// read, 4 adv + 4 ret written).  The recursion runs backwards in time, which a
// lane-per-env kernel can only feed with N/64 waves (64 at N=4096: far too few
// to cover HBM latency), so the roofline variant blocks the TIME axis: a step is
// the affine map x -> delta_t + a_t*x with a_t = gamma*lambda*(1-d_t); maps
// compose associatively, so chunks of TC steps are reduced in parallel to one
// (A, b) pair each (pass 1), a short per-env scan over the chunk pairs gives the
// value entering every chunk (pass 2), and pass 3 replays each chunk from its
// carry-in.  All composition is float64 like the reference's numpy arithmetic.
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

using namespace gymrl;

namespace {

constexpr int kSeqBlock = 64;   // lane-per-env kernels: one wave per workgroup
constexpr int kSeqBatch = 16;   // rows prefetched per batch in the sequential walk
constexpr int kBlkTC    = 16;   // time-chunk length of the blocked variant
constexpr int kBlkV     = 2;    // adjacent envs per lane (8-B row loads, 512 B per wave-row)
constexpr int kBlkBlock = 256;
constexpr int kRedBlock = 256;

// ------------------------------------------------------------------ G1 seq --
// One lane = one env, t = T-1 .. 0 in the reference's operation order:
//   delta = (r + (g*Vn)*(1-d)) - V ;  A = delta + ((g*l)*(1-d))*A
__global__ __launch_bounds__(kSeqBlock) void gae_seq_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const uint8_t* __restrict__ done, const float* __restrict__ next_val, int T, int N,
    double gamma, double gl, float* __restrict__ adv_out, float* __restrict__ ret_out,
    double* __restrict__ partials) {
  const int n = blockIdx.x * kSeqBlock + threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (n < N) {
    double vnext = (double)next_val[n];
    double last = 0.0;
    for (int t1 = T; t1 > 0; t1 -= kSeqBatch) {
      float r[kSeqBatch], v[kSeqBatch];
      uint8_t d[kSeqBatch];
#pragma unroll
      for (int j = 0; j < kSeqBatch; ++j) {
        const int t = t1 - 1 - j;
        if (t >= 0) {
          const size_t o = (size_t)t * N + n;
          r[j] = rew[o]; v[j] = val[o]; d[j] = done[o];
        }
      }
#pragma unroll
      for (int j = 0; j < kSeqBatch; ++j) {
        const int t = t1 - 1 - j;
        if (t >= 0) {
          const double nd = 1.0 - (double)(d[j] != 0);
          const double vv = (double)v[j];
          const double delta = ((double)r[j] + (gamma * vnext) * nd) - vv;
          last = delta + (gl * nd) * last;
          const size_t o = (size_t)t * N + n;
          adv_out[o] = (float)last;
          ret_out[o] = (float)(last + vv);
          s1 += last; s2 += last * last;
          vnext = vv;
        }
      }
    }
  }
  if (partials) {
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s1; partials[2 * blockIdx.x + 1] = s2; }
  }
}

// ------------------------------------------------------------- G1 blocked ---
// Thread = V adjacent envs (one 4V-byte vector load per row) x one chunk of TC
// time steps.  grid.x covers N/V env groups, grid.y covers chunks.
template <int V> struct VecF;
template <> struct VecF<2> { using f = float2; using u = uchar2; };
template <> struct VecF<4> { using f = float4; using u = uchar4; };

template <int TC, int V>
struct Chunk {
  float r[TC][V];
  float v[TC][V];
  uint8_t d[TC][V];
  float vend[V];  // V at (chunk end + 1), or next_val
};

template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&out)[V]) {
  const typename VecF<V>::f t = *reinterpret_cast<const typename VecF<V>::f*>(p);
  out[0] = t.x; out[1] = t.y;
  if constexpr (V == 4) { out[2] = t.z; out[3] = t.w; }
}
template <int V>
__device__ __forceinline__ void ldv(const uint8_t* __restrict__ p, uint8_t (&out)[V]) {
  const typename VecF<V>::u t = *reinterpret_cast<const typename VecF<V>::u*>(p);
  out[0] = t.x; out[1] = t.y;
  if constexpr (V == 4) { out[2] = t.z; out[3] = t.w; }
}

template <int TC, int V>
__device__ __forceinline__ void load_chunk(Chunk<TC, V>& c, const float* __restrict__ rew,
                                           const float* __restrict__ val,
                                           const uint8_t* __restrict__ done,
                                           const float* __restrict__ next_val, int T, int N,
                                           int q, int t0) {
  const int tend = min(t0 + TC, T);
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int t = t0 + j;
    if (t < tend) {
      const size_t o = (size_t)t * N + V * (size_t)q;
      ldv<V>(rew + o, c.r[j]);
      ldv<V>(val + o, c.v[j]);
      ldv<V>(done + o, c.d[j]);
    }
  }
  if (tend == T) ldv<V>(next_val + V * (size_t)q, c.vend);
  else ldv<V>(val + (size_t)tend * N + V * (size_t)q, c.vend);
}

// pass 1: (A, b) of every chunk.  agg layout: [chunk][N] double2.
template <int TC, int V>
__global__ __launch_bounds__(kBlkBlock) void gae_blk_aggregate_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const uint8_t* __restrict__ done, const float* __restrict__ next_val, int T, int N,
    double gamma, double gl, double2* __restrict__ agg) {
  const int q = blockIdx.x * kBlkBlock + threadIdx.x;
  const int c = blockIdx.y;
  if (V * q >= N) return;
  const int t0 = c * TC;
  Chunk<TC, V> ch;
  load_chunk<TC, V>(ch, rew, val, done, next_val, T, N, q, t0);
  const int len = min(TC, T - t0);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    double A = 1.0, b = 0.0;
    double vnext = (double)ch.vend[k];
#pragma unroll
    for (int j = TC - 1; j >= 0; --j) {
      if (j < len) {
        const double nd = 1.0 - (double)(ch.d[j][k] != 0);
        const double vv = (double)ch.v[j][k];
        const double delta = ((double)ch.r[j][k] + (gamma * vnext) * nd) - vv;
        const double a = gl * nd;
        b = delta + a * b;
        A = a * A;
        vnext = vv;
      }
    }
    agg[(size_t)c * N + V * (size_t)q + k] = make_double2(A, b);
  }
}

// pass 2: carry[c][n] = advantage entering chunk c from the future (t = chunk end).
// Latency-bound if walked serially (C dependent HBM round trips), so a workgroup
// takes 64 envs (lanes) x kCarrySeg chunk-segments (waves): every thread loads its
// segment's <= kCarryL aggregates at once, composes them, the segment maps are
// exchanged through LDS, and each thread replays its own segment from its carry-in.
constexpr int kCarrySeg = kGaeCarrySeg;   // 16 (gymrl_device.hpp: gae_carry_scan restates this kernel per lane)
constexpr int kCarryL   = kGaeCarryL;    // chunks per segment per round (registers: 8 double2)

__global__ __launch_bounds__(kSeqBlock * kCarrySeg) void gae_blk_carry_kernel(
    const double2* __restrict__ agg, int C, int N, double* __restrict__ carry) {
  __shared__ double2 seg_map[kCarrySeg][kSeqBlock];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int n = blockIdx.x * kSeqBlock + lane;
  const bool valid = n < N;
  // rounds of kCarrySeg*kCarryL chunks, walked from the last chunk down; `top` = value
  // entering the highest chunk of the round.
  double top = 0.0;
  for (int c_hi = C; c_hi > 0; c_hi -= kCarrySeg * kCarryL) {
    // this thread's chunks: c_hi-1 - seg'*kCarryL - j with seg' = kCarrySeg-1-seg so that
    // seg kCarrySeg-1 holds the highest chunks (matches the compose order below).
    const int c_top = c_hi - 1 - (kCarrySeg - 1 - seg) * kCarryL;
    double2 ab[kCarryL];
#pragma unroll
    for (int j = 0; j < kCarryL; ++j) {
      const int c = c_top - j;
      ab[j] = (valid && c >= 0) ? agg[(size_t)c * N + n] : make_double2(1.0, 0.0);
    }
    double A = 1.0, b = 0.0;   // segment map x -> b + A*x, composed from the top chunk down
#pragma unroll
    for (int j = 0; j < kCarryL; ++j) { b = ab[j].y + ab[j].x * b; A = ab[j].x * A; }
    seg_map[seg][lane] = make_double2(A, b);
    __syncthreads();
    double x = top;
    for (int s2 = kCarrySeg - 1; s2 > seg; --s2) { const double2 m = seg_map[s2][lane]; x = m.y + m.x * x; }
    // value leaving the whole round (needed by the next, lower round)
    double nxt = top;
    for (int s2 = kCarrySeg - 1; s2 >= 0; --s2) { const double2 m = seg_map[s2][lane]; nxt = m.y + m.x * nxt; }
#pragma unroll
    for (int j = 0; j < kCarryL; ++j) {
      const int c = c_top - j;
      if (valid && c >= 0) carry[(size_t)c * N + n] = x;
      x = ab[j].y + ab[j].x * x;
    }
    top = nxt;
    __syncthreads();
  }
}

// pass 3: replay each chunk from its carry-in; write adv/ret; moment partials.
template <int TC, int V>
__global__ __launch_bounds__(kBlkBlock) void gae_blk_apply_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const uint8_t* __restrict__ done, const float* __restrict__ next_val, int T, int N,
    double gamma, double gl, const double* __restrict__ carry, float* __restrict__ adv_out,
    float* __restrict__ ret_out, double* __restrict__ partials) {
  const int q = blockIdx.x * kBlkBlock + threadIdx.x;
  const int c = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  if (V * q < N) {
    const int t0 = c * TC;
    Chunk<TC, V> ch;
    load_chunk<TC, V>(ch, rew, val, done, next_val, T, N, q, t0);
    const int len = min(TC, T - t0);
    double x[V], vnext[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      x[k] = carry[(size_t)c * N + V * (size_t)q + k];
      vnext[k] = (double)ch.vend[k];
    }
#pragma unroll
    for (int j = TC - 1; j >= 0; --j) {
      if (j < len) {
        float a4[V], r4[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const double nd = 1.0 - (double)(ch.d[j][k] != 0);
          const double vv = (double)ch.v[j][k];
          const double delta = ((double)ch.r[j][k] + (gamma * vnext[k]) * nd) - vv;
          x[k] = delta + (gl * nd) * x[k];
          a4[k] = (float)x[k];
          r4[k] = (float)(x[k] + vv);
          s1 += x[k]; s2 += x[k] * x[k];
          vnext[k] = vv;
        }
        const size_t o = (size_t)(t0 + j) * N + V * (size_t)q;
        if constexpr (V == 4) {
          *reinterpret_cast<float4*>(adv_out + o) = make_float4(a4[0], a4[1], a4[2], a4[3]);
          *reinterpret_cast<float4*>(ret_out + o) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        } else {
          *reinterpret_cast<float2*>(adv_out + o) = make_float2(a4[0], a4[1]);
          *reinterpret_cast<float2*>(ret_out + o) = make_float2(r4[0], r4[1]);
        }
      }
    }
  }
  if (partials) {
    __shared__ double sm[2][kBlkBlock / 64];
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wid] = s1; sm[1][wid] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int w = 0; w < kBlkBlock / 64; ++w) { a += sm[0][w]; b += sm[1][w]; }
      const size_t bid = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
      partials[2 * bid] = a; partials[2 * bid + 1] = b;
    }
  }
}

// pass 1 folded into the producer: the rollout composes each step into its chunk's (A, b)
// as soon as V_{t+1} is known (the categorical-sample kernel of step t+1 calls this body),
// so the HBM-bound GAE launch no longer re-reads r, v, d (variant 2: carry + apply only).
// Forward composition F <- F o f_t of x -> delta_t + a_t x:  b += A*delta_t ; A *= a_t.
__global__ __launch_bounds__(kRedBlock) void gae_online_kernel(
    const float* __restrict__ rew_prev, const uint8_t* __restrict__ done_prev,
    const float* __restrict__ val_prev, const float* __restrict__ val_cur, int N, double gamma,
    double gl, int first, int last, double* __restrict__ running, double2* __restrict__ agg_row,
    double gl2, double* __restrict__ running2, double2* __restrict__ agg_row2) {
  const int i = blockIdx.x * kRedBlock + threadIdx.x;
  if (i >= N) return;
  gymrl::gae_online_compose(rew_prev[i], done_prev[i], val_prev[i], val_cur[i], gamma, gl, first, last,
                            running, agg_row, N, i);
  if (running2)
    gymrl::gae_online_compose(rew_prev[i], done_prev[i], val_prev[i], val_cur[i], gamma, gl2, first, last,
                              running2, agg_row2, N, i);
}

// Deterministic final reduction of (s1, s2) partials -> moments (count, sum, sumsq).
__global__ __launch_bounds__(kRedBlock) void moments_finalize_kernel(
    const double* __restrict__ partials, int nparts, double count,
    double* __restrict__ moments_out) {
  __shared__ double sm[2][kRedBlock];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kRedBlock) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = kRedBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { moments_out[0] = count; moments_out[1] = sm[0][0]; moments_out[2] = sm[1][0]; }
}

// ------------------------------------------------------------------ G2 ------
// utils/buffer.py:21-35: float32 throughout; delta uses dw, recursion uses done.
__global__ __launch_bounds__(kSeqBlock) void gae_dw_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const float* __restrict__ next_val, const uint8_t* __restrict__ done,
    const uint8_t* __restrict__ dw, int T, int N, float gamma, float gl,
    float* __restrict__ adv_out, float* __restrict__ vt_out, double* __restrict__ partials) {
  const int n = blockIdx.x * kSeqBlock + threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (n < N) {
    float gae = 0.0f;
    for (int t1 = T; t1 > 0; t1 -= kSeqBatch) {
      float r[kSeqBatch], v[kSeqBatch], vn[kSeqBatch];
      uint8_t d[kSeqBatch], w[kSeqBatch];
#pragma unroll
      for (int j = 0; j < kSeqBatch; ++j) {
        const int t = t1 - 1 - j;
        if (t >= 0) {
          const size_t o = (size_t)t * N + n;
          r[j] = rew[o]; v[j] = val[o]; vn[j] = next_val[o]; d[j] = done[o]; w[j] = dw[o];
        }
      }
#pragma unroll
      for (int j = 0; j < kSeqBatch; ++j) {
        const int t = t1 - 1 - j;
        if (t >= 0) {
          const float delta = (r[j] + (gamma * vn[j]) * (1.0f - (float)(w[j] != 0))) - v[j];
          gae = (gl * gae) * (1.0f - (float)(d[j] != 0)) + delta;
          const size_t o = (size_t)t * N + n;
          adv_out[o] = gae;
          vt_out[o] = gae + v[j];
          s1 += (double)gae; s2 += (double)gae * (double)gae;
        }
      }
    }
  }
  if (partials) {
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s1; partials[2 * blockIdx.x + 1] = s2; }
  }
}

// ------------------------------------------------------------------ G3 ------
__global__ __launch_bounds__(kSeqBlock) void gae_decoupled_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const uint8_t* __restrict__ done, const float* __restrict__ next_val, int T, int N,
    double gamma, double gla, double glc, float* __restrict__ adv_out,
    float* __restrict__ ret_out) {
  const int n = blockIdx.x * kSeqBlock + threadIdx.x;
  if (n >= N) return;
  double vnext = (double)next_val[n];
  double la = 0.0, lc = 0.0;
  for (int t1 = T; t1 > 0; t1 -= kSeqBatch) {
    float r[kSeqBatch], v[kSeqBatch];
    uint8_t d[kSeqBatch];
#pragma unroll
    for (int j = 0; j < kSeqBatch; ++j) {
      const int t = t1 - 1 - j;
      if (t >= 0) {
        const size_t o = (size_t)t * N + n;
        r[j] = rew[o]; v[j] = val[o]; d[j] = done[o];
      }
    }
#pragma unroll
    for (int j = 0; j < kSeqBatch; ++j) {
      const int t = t1 - 1 - j;
      if (t >= 0) {
        const double nd = 1.0 - (double)(d[j] != 0);
        const double vv = (double)v[j];
        const double delta = ((double)r[j] + (gamma * vnext) * nd) - vv;
        la = delta + (gla * nd) * la;
        lc = delta + (glc * nd) * lc;
        const size_t o = (size_t)t * N + n;
        adv_out[o] = (float)la;
        ret_out[o] = (float)(lc + vv);
        vnext = vv;
      }
    }
  }
}

// G3 blocked: the G1 machinery with two affine maps per chunk (same delta, decay factors gla / glc).
// agg / carry layout: [actor: C][N] then [critic: C][N].
template <int TC, int V>
__global__ __launch_bounds__(kBlkBlock) void gae2_blk_aggregate_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const uint8_t* __restrict__ done, const float* __restrict__ next_val, int T, int N,
    double gamma, double gla, double glc, double2* __restrict__ agg_a, double2* __restrict__ agg_c) {
  const int q = blockIdx.x * kBlkBlock + threadIdx.x;
  const int c = blockIdx.y;
  if (V * q >= N) return;
  const int t0 = c * TC;
  Chunk<TC, V> ch;
  load_chunk<TC, V>(ch, rew, val, done, next_val, T, N, q, t0);
  const int len = min(TC, T - t0);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    double Aa = 1.0, ba = 0.0, Ac = 1.0, bc = 0.0;
    double vnext = (double)ch.vend[k];
#pragma unroll
    for (int j = TC - 1; j >= 0; --j) {
      if (j < len) {
        const double nd = 1.0 - (double)(ch.d[j][k] != 0);
        const double vv = (double)ch.v[j][k];
        const double delta = ((double)ch.r[j][k] + (gamma * vnext) * nd) - vv;
        const double a = gla * nd, cc = glc * nd;
        ba = delta + a * ba;  Aa = a * Aa;
        bc = delta + cc * bc; Ac = cc * Ac;
        vnext = vv;
      }
    }
    agg_a[(size_t)c * N + V * (size_t)q + k] = make_double2(Aa, ba);
    agg_c[(size_t)c * N + V * (size_t)q + k] = make_double2(Ac, bc);
  }
}

template <int TC, int V>
__global__ __launch_bounds__(kBlkBlock) void gae2_blk_apply_kernel(
    const float* __restrict__ rew, const float* __restrict__ val,
    const uint8_t* __restrict__ done, const float* __restrict__ next_val, int T, int N,
    double gamma, double gla, double glc, const double* __restrict__ carry_a, const double* __restrict__ carry_c,
    float* __restrict__ adv_out, float* __restrict__ ret_out) {
  const int q = blockIdx.x * kBlkBlock + threadIdx.x;
  const int c = blockIdx.y;
  if (V * q >= N) return;
  const int t0 = c * TC;
  Chunk<TC, V> ch;
  load_chunk<TC, V>(ch, rew, val, done, next_val, T, N, q, t0);
  const int len = min(TC, T - t0);
  double xa[V], xc[V], vnext[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    xa[k] = carry_a[(size_t)c * N + V * (size_t)q + k];
    xc[k] = carry_c[(size_t)c * N + V * (size_t)q + k];
    vnext[k] = (double)ch.vend[k];
  }
#pragma unroll
  for (int j = TC - 1; j >= 0; --j) {
    if (j < len) {
      float a4[V], r4[V];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const double nd = 1.0 - (double)(ch.d[j][k] != 0);
        const double vv = (double)ch.v[j][k];
        const double delta = ((double)ch.r[j][k] + (gamma * vnext[k]) * nd) - vv;
        xa[k] = delta + (gla * nd) * xa[k];
        xc[k] = delta + (glc * nd) * xc[k];
        a4[k] = (float)xa[k];
        r4[k] = (float)(xc[k] + vv);
        vnext[k] = vv;
      }
      const size_t o = (size_t)(t0 + j) * N + V * (size_t)q;
      *reinterpret_cast<float2*>(adv_out + o) = make_float2(a4[0], a4[1]);
      *reinterpret_cast<float2*>(ret_out + o) = make_float2(r4[0], r4[1]);
    }
  }
}

// ------------------------------------------------------- moments / normalise
__global__ __launch_bounds__(kRedBlock) void moments_partial_kernel(
    const float* __restrict__ x, int64_t n, double* __restrict__ partials) {
  double s1 = 0.0, s2 = 0.0;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kRedBlock;
  for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const double v = (double)x[(n4 << 2) + threadIdx.x];
    s1 += v; s2 += v * v;
  }
  __shared__ double sm[2][kRedBlock / 64];
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wid] = s1; sm[1][wid] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int w = 0; w < kRedBlock / 64; ++w) { a += sm[0][w]; b += sm[1][w]; }
    partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b;
  }
}

__global__ __launch_bounds__(kRedBlock) void normalize_kernel(
    float* __restrict__ x, int64_t n, const double* __restrict__ moments, int ddof, double eps) {
  const double cnt = moments[0];
  const double mean = moments[1] / cnt;
  double var = (moments[2] - cnt * mean * mean) / (cnt - (double)ddof);
  var = var > 0.0 ? var : 0.0;
  const double denom = sqrt(var) + eps;
  const int64_t stride = (int64_t)gridDim.x * kRedBlock;
  for (int64_t i = (int64_t)blockIdx.x * kRedBlock + threadIdx.x; i < n; i += stride)
    x[i] = (float)(((double)x[i] - mean) / denom);
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

size_t gymrl_gae_workspace_bytes(int T, int N) {
  if (T <= 0 || N <= 0) return 0;
  const size_t C = (size_t)cdiv(T, kBlkTC);
  const size_t agg = C * (size_t)N * sizeof(double2);
  const size_t carry = C * (size_t)N * sizeof(double);
  const size_t parts = sizeof(double) * 2 * ((size_t)cdiv(cdiv(N, kBlkV), kBlkBlock) * C + (size_t)cdiv(N, kSeqBlock) + 16);
  return agg + carry + parts + 1024;
}

int gymrl_gae(const float* rew, const float* val, const uint8_t* done, const float* next_val,
              int T, int N, double gamma, double lam, float* adv_out, float* ret_out,
              double* moments_out, int variant, void* workspace, void* stream_) {
  if (!rew || !val || !done || !next_val || !adv_out || !ret_out || T < 0 || N < 0) return -22;
  if (T == 0 || N == 0) return 0;
  if (moments_out && !workspace) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  // ppo_lunarlander.py:192: `gamma * gae_lambda * (1 - dones[t])` multiplies a python
  // float by a numpy float32 scalar, which (NumPy >= 2, NEP 50) is a float32 product:
  // the decay factor is float32(gamma*lambda); everything else stays float64.
  const double gl = (double)(float)(gamma * lam);
  const bool vec_ok = (N % 4 == 0) && aligned16(rew) && aligned16(val) && aligned16(adv_out) &&
                      aligned16(ret_out) && aligned16(next_val) &&
                      ((reinterpret_cast<uintptr_t>(done) & 3) == 0);
  if (variant == 3 && !(vec_ok && workspace)) return -22;   // the rollout only composes maps / carries under the same layout rules
  if ((variant == 1 || variant == 2 || variant == 3) && vec_ok && workspace) {
    const int C = cdiv(T, kBlkTC);
    char* ws = (char*)workspace;
    double2* agg = (double2*)ws;
    double* carry = (double*)(ws + (size_t)C * N * sizeof(double2));
    double* parts = carry + (size_t)C * N;
    dim3 grid(cdiv(N / kBlkV, kBlkBlock), C);
    if (variant == 1)   // variant 2: the chunk maps were composed online during the rollout
      hipLaunchKernelGGL((gae_blk_aggregate_kernel<kBlkTC, kBlkV>), grid, dim3(kBlkBlock), 0, stream, rew, val, done,
                         next_val, T, N, gamma, gl, agg);
    if (variant != 3)   // variant 3: the persistent rollout ran this pass for its own envs at its tail (gymrl_device.hpp gae_carry_scan)
      hipLaunchKernelGGL(gae_blk_carry_kernel, dim3(cdiv(N, kSeqBlock)), dim3(kSeqBlock * kCarrySeg), 0,
                         stream, agg, C, N, carry);
    hipLaunchKernelGGL((gae_blk_apply_kernel<kBlkTC, kBlkV>), grid, dim3(kBlkBlock), 0, stream, rew, val, done,
                       next_val, T, N, gamma, gl, carry, adv_out, ret_out,
                       moments_out ? parts : nullptr);
    if (moments_out)
      hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(kRedBlock), 0, stream, parts,
                         (int)(grid.x * grid.y), (double)T * (double)N, moments_out);
  } else {
    double* parts = moments_out ? (double*)workspace : nullptr;
    const int nb = cdiv(N, kSeqBlock);
    hipLaunchKernelGGL(gae_seq_kernel, dim3(nb), dim3(kSeqBlock), 0, stream, rew, val, done,
                       next_val, T, N, gamma, gl, adv_out, ret_out, parts);
    if (moments_out)
      hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(kRedBlock), 0, stream, parts, nb,
                         (double)T * (double)N, moments_out);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_gae_chunk(void) { return kBlkTC; }

int gymrl_gae_online_flush(const gymrl_gae_online* o, const float* val_cur, int N, void* stream_) {
  if (!o || !val_cur || !o->rew_prev || !o->done_prev || !o->val_prev || !o->running || !o->gae_workspace ||
      N <= 0 || o->t_prev < 0 || o->t_prev >= o->T)
    return -22;
  const int c = o->t_prev / kBlkTC;
  const int first = (o->t_prev % kBlkTC) == 0, last = (o->t_prev % kBlkTC) == kBlkTC - 1 || o->t_prev == o->T - 1;
  double2* agg = (double2*)o->gae_workspace + (size_t)c * N;
  const bool two = o->lam2 > 0.0 && o->running2;
  const size_t C = (size_t)cdiv(o->T, kBlkTC);
  hipLaunchKernelGGL(gae_online_kernel, dim3(cdiv(N, kRedBlock)), dim3(kRedBlock), 0, (hipStream_t)stream_,
                     o->rew_prev, o->done_prev, o->val_prev, val_cur, N, o->gamma,
                     two ? o->gamma * o->lam : (double)(float)(o->gamma * o->lam), first, last, o->running, agg,
                     o->gamma * o->lam2, two ? o->running2 : (double*)nullptr, two ? agg + C * (size_t)N : (double2*)nullptr);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_gae_dw(const float* rew, const float* val, const float* next_val, const uint8_t* done,
                 const uint8_t* dw, int T, int N, double gamma, double lam, float* adv_out,
                 float* vtarget_out, double* moments_out, void* workspace, void* stream_) {
  if (!rew || !val || !next_val || !done || !dw || !adv_out || !vtarget_out || T < 0 || N < 0) return -22;
  if (T == 0 || N == 0) return 0;
  if (moments_out && !workspace) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = cdiv(N, kSeqBlock);
  double* parts = moments_out ? (double*)workspace : nullptr;
  hipLaunchKernelGGL(gae_dw_kernel, dim3(nb), dim3(kSeqBlock), 0, stream, rew, val, next_val, done,
                     dw, T, N, (float)gamma, (float)(gamma * lam), adv_out, vtarget_out, parts);
  if (moments_out)
    hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(kRedBlock), 0, stream, parts, nb,
                       (double)T * (double)N, moments_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

size_t gymrl_gae_decoupled_workspace_bytes(int T, int N) {
  if (T <= 0 || N <= 0) return 0;
  const size_t C = (size_t)cdiv(T, kBlkTC);
  return 2 * C * (size_t)N * (sizeof(double2) + sizeof(double)) + 1024;
}

int gymrl_gae_decoupled(const float* rew, const float* val, const uint8_t* done,
                        const float* next_val, int T, int N, double gamma, double lam_actor,
                        double lam_critic, float* adv_actor_out, float* ret_out, int variant, void* workspace,
                        void* stream_) {
  if (!rew || !val || !done || !next_val || !adv_actor_out || !ret_out || T < 0 || N < 0) return -22;
  if (T == 0 || N == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  const double gla = gamma * lam_actor, glc = gamma * lam_critic;
  const bool vec_ok = (N % 4 == 0) && aligned16(rew) && aligned16(val) && aligned16(adv_actor_out) &&
                      aligned16(ret_out) && aligned16(next_val) &&
                      ((reinterpret_cast<uintptr_t>(done) & 3) == 0);
  if (variant == 3 && !(vec_ok && workspace)) return -22;   // the rollout only composes maps / carries under the same layout rules
  if ((variant == 1 || variant == 2 || variant == 3) && vec_ok && workspace) {
    const int C = cdiv(T, kBlkTC);
    const size_t CN = (size_t)C * N;
    double2* agg_a = (double2*)workspace;
    double2* agg_c = agg_a + CN;
    double* carry_a = (double*)(agg_c + CN);
    double* carry_c = carry_a + CN;
    dim3 grid(cdiv(N / kBlkV, kBlkBlock), C);
    if (variant == 1)
      hipLaunchKernelGGL((gae2_blk_aggregate_kernel<kBlkTC, kBlkV>), grid, dim3(kBlkBlock), 0, stream, rew, val, done,
                         next_val, T, N, gamma, gla, glc, agg_a, agg_c);
    hipLaunchKernelGGL(gae_blk_carry_kernel, dim3(cdiv(N, kSeqBlock)), dim3(kSeqBlock * kCarrySeg), 0, stream, agg_a, C, N,
                       carry_a);
    hipLaunchKernelGGL(gae_blk_carry_kernel, dim3(cdiv(N, kSeqBlock)), dim3(kSeqBlock * kCarrySeg), 0, stream, agg_c, C, N,
                       carry_c);
    hipLaunchKernelGGL((gae2_blk_apply_kernel<kBlkTC, kBlkV>), grid, dim3(kBlkBlock), 0, stream, rew, val, done, next_val,
                       T, N, gamma, gla, glc, carry_a, carry_c, adv_actor_out, ret_out);
  } else {
    hipLaunchKernelGGL(gae_decoupled_kernel, dim3(cdiv(N, kSeqBlock)), dim3(kSeqBlock), 0, stream, rew, val, done,
                       next_val, T, N, gamma, gla, glc, adv_actor_out, ret_out);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_moments(const float* x, int64_t n, double* moments_out, void* workspace, void* stream_) {
  if (!x || !moments_out || !workspace || n < 0) return -22;
  double* parts = (double*)workspace;   // >= gymrl_reduce_workspace_bytes()
  hipStream_t stream = (hipStream_t)stream_;
  int nb = cdiv(n > 0 ? n : 1, (int64_t)kRedBlock * 16);
  if (nb > 2048) nb = 2048;
  if (!aligned16(x)) return -22;
  hipLaunchKernelGGL(moments_partial_kernel, dim3(nb), dim3(kRedBlock), 0, stream, x, n, parts);
  hipLaunchKernelGGL(moments_finalize_kernel, dim3(1), dim3(kRedBlock), 0, stream, parts, nb,
                     (double)n, moments_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_normalize(float* x, int64_t n, const double* moments, int ddof, double eps,
                    void* stream_) {
  if (!x || !moments || n < 0 || ddof < 0) return -22;
  if (n == 0) return 0;
  int nb = cdiv(n, (int64_t)kRedBlock * 8);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(kRedBlock), 0, (hipStream_t)stream_, x, n,
                     moments, ddof, eps);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
