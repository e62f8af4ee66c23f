// lin.hip — low-latency Linear(+activation) layers for the off-policy networks' update and acting path.
//
// The reference's DQN / Rainbow / SAC / TD3 / DDPG updates (dqn_cartpole.py:164-196, rainbow_dqn_cartpole.py:311-361,
// sac_pendulum.py:213-267, td3_pendulum.py:204-250) push 64-256 rows through 256-wide torch Linear layers; vectorised,
// the acting forward runs on 4096-8192 rows.  At these sizes a layer is a few MFLOP: a library GEMM plus its bias
// add, activation, activation-gradient, bias-gradient and gradient-copy launches is 3-6 launches of 3-5 us each per
// layer and direction, and a SAC update was 187 launches (`profiles/r01_sac_graph_kernel_stats.csv`).  Here a layer is
// ONE launch per direction:
//
//   lin_fwd         Y  = act(X W^T + b)                       (X may be the concatenation [X | X2] of two row blocks)
//   lin_bwd_input   dX = (dY * act'(Y)) W                     (split back into dX | dX2; either may be skipped)
//   lin_bwd_weight  dW (+)= (dY * act'(Y))^T X,  db (+)= column sums of dY * act'(Y)
//
// and up to GYMRL_LIN_MAX_ITEMS independent layers of one shape (the twin Q networks, the actor's mean / log_std heads,
// online + target networks) share a launch (blockIdx.y).
//
// Mapping: one wavefront owns a 16 x 16 (NT = 1) or 16 x 64 (NT = 4) output tile and walks the whole reduction with
// v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation, 8 passes); operands come straight from global memory
// (everything is L2-resident: a 256 x 256 weight is 256 KiB) — 16-byte loads along the reduction where it is the
// contiguous dimension (lin_fwd: both operands; lin_bwd_input: dY and Y), dword loads of 64-byte row segments otherwise.
// No LDS, no barriers: with 128-rows x 256 outputs = 128 independent waves a launch lasts one wave's ~2 us chain.
// Tiles are 16-row granular so ragged B / N / K are handled by predicated loads (zeros) and stores.
//
// Accumulation order (documented for the tests; all sums are f32 fma chains inside the MFMA unit):
//   lin_fwd:        for k0 = 0, 16, ...: for e = 0..3: the MFMA adds the four products k = k0 + 4 q + e, q = 0..3 in order
//   lin_bwd_input:  the same with n in place of k
//   lin_bwd_weight: for b0 = lo, lo + 16, ...: for e = 0..3: rows b = b0 + 4 e + q, q = 0..3; B > 512 rows are cut into
//                   slices whose partial tiles are added in slice order by a second launch (fixed order, no atomics)
#include "lin_device.hpp"

namespace {

using namespace gymrl;

constexpr int kItems = GYMRL_LIN_MAX_ITEMS;
constexpr int kWavesPerBlock = 4;
constexpr int kChunk = 8;       // reduction steps (of 16) whose loads are in flight together

using lin::mfma16;
using lin::act_fwd;
using lin::act_bwd;

struct LinFwd {
  const float* X[kItems]; const float* X2[kItems]; const float* W[kItems]; const float* b[kItems]; float* Y[kItems];
  int32_t* argmax[kItems];
  int act[kItems]; float lo[kItems], hi[kItems];
  int B, K, K1, N, ldx, ldx2, ldy, col_groups;
};

template <int NT, bool VEC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void lin_fwd_kernel(const LinFwd a) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int item = blockIdx.y;
  const int rt = wave / a.col_groups, cg = wave - rt * a.col_groups;
  if (rt * 16 >= a.B) return;
  const float* __restrict__ X = a.X[item];
  const float* __restrict__ X2 = a.X2[item];
  const float* __restrict__ W = a.W[item];
  const int act = a.act[item];
  const float lo = a.lo[item], hi = a.hi[item];
  const int row = rt * 16 + r;
  const bool row_ok = row < a.B;
  const int nb = cg * 16 * NT;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = zero;
  const float* xrow = X + (size_t)(row_ok ? row : 0) * a.ldx;
  const float* x2row = X2 ? X2 + (size_t)(row_ok ? row : 0) * a.ldx2 : nullptr;
  if constexpr (VEC) {
    // all of a chunk's loads (unconditional, clamped addresses) are in flight before its first MFMA
    const float* wrow[NT];
    bool n_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = nb + 16 * t + r;
      n_ok[t] = n < a.N;
      wrow[t] = W + (size_t)(n_ok[t] ? n : 0) * a.K;
    }
    for (int kc0 = 0; kc0 < a.K; kc0 += 16 * kChunk) {
      f32x4 xa[kChunk], wb[kChunk][NT];
#pragma unroll
      for (int c = 0; c < kChunk; ++c) {
        if (kc0 + 16 * c >= a.K) break;                     // (wave-uniform) short reductions: K = 4 is one step, not eight
        const int k = kc0 + 16 * c + 4 * q;
        const int ks = k < a.K ? k : 0;
        xa[c] = *reinterpret_cast<const f32x4*>(xrow + ks);
#pragma unroll
        for (int t = 0; t < NT; ++t) wb[c][t] = *reinterpret_cast<const f32x4*>(wrow[t] + ks);
      }
#pragma unroll
      for (int c = 0; c < kChunk; ++c) {
        if (kc0 + 16 * c >= a.K) break;
        const bool k_ok = kc0 + 16 * c + 4 * q < a.K;
        const f32x4 x = (row_ok && k_ok) ? xa[c] : zero;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 w = (n_ok[t] && k_ok) ? wb[c][t] : zero;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t] = mfma16(x[e], w[e], acc[t]);
        }
      }
    }
  } else {
    for (int k0 = 0; k0 < a.K; k0 += 16) {
      const int k = k0 + 4 * q;
      f32x4 xa, wb[NT];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k + e;
        float v = 0.0f;
        if (row_ok && kk < a.K) v = kk < a.K1 ? xrow[kk] : x2row[kk - a.K1];
        xa[e] = v;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = nb + 16 * t + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) wb[t][e] = (n < a.N && k + e < a.K) ? W[(size_t)n * a.K + k + e] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(xa[e], wb[t][e], acc[t]);
    }
  }
  const float* __restrict__ bias = a.b[item];
  float* __restrict__ Y = a.Y[item];
  if (act == GYMRL_ACT_DUELING) {
    // columns 0..A-1 = advantage stream, column A = value stream (N = A + 1 <= 16: all in this wave's first tile):
    // q = value + advantage - mean(advantage) (rainbow_dqn_cartpole.py:112), written as [B, A]
    if (cg != 0) return;
    const int A = a.N - 1;
    const float bv = (bias && r < a.N) ? bias[r] : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float z = acc[0][g] + bv;
      float s = r < A ? z : 0.0f;
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
      const float v = __shfl(z, (lane & 48) | A, 64);
      const int ro = rt * 16 + 4 * q + g;
      const float qv = v + (z - s / (float)A);
      if (r < A && ro < a.B) Y[(size_t)ro * a.ldy + r] = qv;
      if (a.argmax[item]) {                        // greedy action: first index of the maximum
        float best = r < A ? qv : -3.4e38f;
        int bi = r < A ? r : 16;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          const float ov = __shfl_xor(best, m, 64);
          const int oi = __shfl_xor(bi, m, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (r == 0 && ro < a.B) a.argmax[item][ro] = bi;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = nb + 16 * t + r;
    if (n >= a.N) continue;
    const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ro = rt * 16 + 4 * q + g;
      if (ro < a.B) Y[(size_t)ro * a.ldy + n] = act_fwd(acc[t][g] + bv, act, lo, hi);
    }
  }
}

struct LinBwdIn {
  const float* dY[kItems]; const float* Y[kItems]; const float* W[kItems]; float* dX[kItems]; float* dX2[kItems];
  int act[kItems]; float lo[kItems], hi[kItems];
  int B, N, K, K1, ldy, lddx, lddx2, accumulate, col_groups, n_sum;
};

template <int NT, bool VEC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void lin_bwd_input_kernel(const LinBwdIn a) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int item = blockIdx.y;
  const int rt = wave / a.col_groups, cg = wave - rt * a.col_groups;
  if (rt * 16 >= a.B) return;
  const int row = rt * 16 + r;
  const bool row_ok = row < a.B;
  const int kb = cg * 16 * NT;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = zero;
  const size_t roff = (size_t)(row_ok ? row : 0) * a.ldy;
  // n_sum > 1: the items' products are summed (layers fed by ONE input: the gradient of that input) — the
  // reduction simply runs on over the next item's dY / W
  for (int it = item; it < item + a.n_sum; ++it) {
  const float* __restrict__ dY = a.dY[it];
  const float* __restrict__ Yv = a.Y[it];
  const float* __restrict__ W = a.W[it];
  const int act = a.act[it];
  const float lo = a.lo[it], hi = a.hi[it];
  if constexpr (VEC) {
    int kcol[NT];
    bool k_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int kc = kb + 16 * t + r;
      k_ok[t] = kc < a.K;
      kcol[t] = k_ok[t] ? kc : 0;
    }
    for (int nc0 = 0; nc0 < a.N; nc0 += 16 * kChunk) {
      f32x4 dy[kChunk], yv[kChunk], wb[kChunk][NT];
#pragma unroll
      for (int c = 0; c < kChunk; ++c) {
        if (nc0 + 16 * c >= a.N) break;
        const int n = nc0 + 16 * c + 4 * q;
        const int ns = n < a.N ? n : 0;
        dy[c] = *reinterpret_cast<const f32x4*>(dY + roff + ns);
        if (act != GYMRL_ACT_NONE) yv[c] = *reinterpret_cast<const f32x4*>(Yv + roff + ns);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) wb[c][t][e] = W[(size_t)(ns + e) * a.K + kcol[t]];
      }
#pragma unroll
      for (int c = 0; c < kChunk; ++c) {
        if (nc0 + 16 * c >= a.N) break;
        const bool ok = row_ok && nc0 + 16 * c + 4 * q < a.N;
        f32x4 dz = ok ? dy[c] : zero;
        if (act != GYMRL_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dz[e] *= act_bwd(yv[c][e], act, lo, hi);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t] = mfma16(dz[e], k_ok[t] ? wb[c][t][e] : 0.0f, acc[t]);
      }
    }
  } else {
    for (int n0 = 0; n0 < a.N; n0 += 16) {
      const int n = n0 + 4 * q;
      f32x4 dz;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.0f;
        if (row_ok && n + e < a.N) {
          v = dY[roff + n + e];
          if (act != GYMRL_ACT_NONE) v *= act_bwd(Yv[roff + n + e], act, lo, hi);
        }
        dz[e] = v;
      }
      f32x4 wb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int kc = kb + 16 * t + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) wb[t][e] = (n + e < a.N && kc < a.K) ? W[(size_t)(n + e) * a.K + kc] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(dz[e], wb[t][e], acc[t]);
    }
  }
  }
  float* __restrict__ dX = a.dX[item];
  float* __restrict__ dX2 = a.dX2[item];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kc = kb + 16 * t + r;
    if (kc >= a.K) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ro = rt * 16 + 4 * q + g;
      if (ro >= a.B) continue;
      float* dst = kc < a.K1 ? (dX ? dX + (size_t)ro * a.lddx + kc : nullptr)
                             : (dX2 ? dX2 + (size_t)ro * a.lddx2 + (kc - a.K1) : nullptr);
      if (dst) *dst = a.accumulate ? *dst + acc[t][g] : acc[t][g];
    }
  }
}

struct LinBwdW {
  const float* dY[kItems]; const float* Y[kItems]; const float* X[kItems]; const float* X2[kItems];
  float* dW[kItems]; float* db[kItems];
  float* partial;              // slices > 1: [item][slice][N*K + N]
  int act[kItems]; float lo[kItems], hi[kItems];
  int B, N, K, K1, ldy, ldx, ldx2, accumulate, col_groups, slices, rows_per_slice;
};

template <int NT>
__global__ __launch_bounds__(64 * kWavesPerBlock) void lin_bwd_weight_kernel(const LinBwdW a) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int item = blockIdx.y, slice = blockIdx.z;
  const int nt = wave / a.col_groups, cg = wave - nt * a.col_groups;
  if (nt * 16 >= a.N) return;
  const float* __restrict__ dY = a.dY[item];
  const float* __restrict__ Yv = a.Y[item];
  const float* __restrict__ X = a.X[item];
  const float* __restrict__ X2 = a.X2[item];
  const int act = a.act[item];
  const float lo = a.lo[item], hi = a.hi[item];
  const int n = nt * 16 + r;
  const bool n_ok = n < a.N;
  const int kb = cg * 16 * NT;
  const int b_lo = slice * a.rows_per_slice;
  const int b_hi = min(a.B, b_lo + a.rows_per_slice);
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = zero;
  float colsum = 0.0f;
  // chunks of kChunk 16-row steps: every load of a chunk (unconditional, clamped addresses) is issued before its first MFMA
  const int ns = n_ok ? n : 0;
  for (int bc0 = b_lo; bc0 < b_hi; bc0 += 16 * kChunk) {
    float dy[kChunk][4], yv[kChunk][4], xb[kChunk][NT][4];
#pragma unroll
    for (int c = 0; c < kChunk; ++c) {
      if (bc0 + 16 * c >= b_hi) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int b = bc0 + 16 * c + 4 * e + q;
        const size_t bs = (size_t)(b < b_hi ? b : b_lo);
        dy[c][e] = dY[bs * a.ldy + ns];
        if (act != GYMRL_ACT_NONE) yv[c][e] = Yv[bs * a.ldy + ns];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int kc = kb + 16 * t + r;
          const int ks = kc < a.K ? kc : 0;
          const float* px = ks < a.K1 ? X + bs * a.ldx + ks : X2 + bs * a.ldx2 + (ks - a.K1);
          xb[c][t][e] = *px;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kChunk; ++c) {
      if (bc0 + 16 * c >= b_hi) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = n_ok && bc0 + 16 * c + 4 * e + q < b_hi;
        float dz = ok ? dy[c][e] : 0.0f;
        if (act != GYMRL_ACT_NONE) dz *= ok ? act_bwd(yv[c][e], act, lo, hi) : 0.0f;
        colsum += dz;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(dz, kb + 16 * t + r < a.K ? xb[c][t][e] : 0.0f, acc[t]);
      }
    }
  }
  // bias gradient: the four row groups' serial sums, added pairwise (q0 + q1) + (q2 + q3)
  colsum += __shfl_xor(colsum, 16, 64);
  colsum += __shfl_xor(colsum, 32, 64);
  const size_t wk = (size_t)a.N * a.K;
  float* __restrict__ dW = a.dW[item];
  float* __restrict__ db = a.db[item];
  float* part = a.slices > 1 ? a.partial + ((size_t)item * a.slices + slice) * (wk + a.N) : nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kc = kb + 16 * t + r;
    if (kc >= a.K) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int no = nt * 16 + 4 * q + g;
      if (no >= a.N) continue;
      const size_t o = (size_t)no * a.K + kc;
      if (part) part[o] = acc[t][g];
      else dW[o] = a.accumulate ? dW[o] + acc[t][g] : acc[t][g];
    }
  }
  if (cg == 0 && q == 0 && n_ok) {
    if (part) part[wk + n] = colsum;
    else if (db) db[n] = a.accumulate ? db[n] + colsum : colsum;
  }
}

// Large batches (>= 16384 rows), N and K multiples of 64, no activation (PPO-full's 128- and 256-wide layers at 262144-row
// micro-batches).  The 16 x 16 tiles above re-read dY and X once per tile pair — at N = K = 128 eight times the operands
// through L2, 1 GB per launch at 131072 rows: 189 us.  Here a wave owns a 64 x 64 block of dW for one row slice: lane r of
// a quarter-wave holds columns 4r .. 4r + 3 of the block (sub-tile a = column 4r + a), so ONE 16-byte load of dY and one
// of X per lane and 4-row step feed 16 MFMAs, and the next 16 rows are in flight while the current 16 multiply.
// acc[a][t][g] = dW[n0 + 4 (4q + g) + a][k0 + 4r + t].
constexpr int kBigSteps = 4;     // 4-row MFMA steps per buffer (16 rows)
__global__ __launch_bounds__(64 * kWavesPerBlock) void lin_bwd_weight_big_kernel(const LinBwdW a) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int item = blockIdx.y, slice = blockIdx.z;
  const int kgroups = a.K >> 6;
  const int ng = wave / kgroups, kg = wave - ng * kgroups;
  if (ng * 64 >= a.N) return;
  const int n0 = ng * 64, k0 = kg * 64;
  const float* __restrict__ dY = a.dY[item] + n0 + 4 * r;
  const float* __restrict__ X = a.X[item] + k0 + 4 * r;
  const int b_lo = slice * a.rows_per_slice;
  const int b_hi = min(a.B, b_lo + a.rows_per_slice);
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = zero;
  f32x4 colsum = zero;
  f32x4 dyA[kBigSteps], xA[kBigSteps], dyB[kBigSteps], xB[kBigSteps];
  auto load = [&](f32x4* dy, f32x4* x, int bc0) {
#pragma unroll
    for (int s = 0; s < kBigSteps; ++s) {
      const int bb = bc0 + 4 * s + q;
      const size_t bs = (size_t)(bb < b_hi ? bb : b_lo);
      dy[s] = *reinterpret_cast<const f32x4*>(dY + bs * a.ldy);
      x[s] = *reinterpret_cast<const f32x4*>(X + bs * a.ldx);
    }
  };
  auto mul = [&](const f32x4* dy, const f32x4* x, int bc0) {
#pragma unroll
    for (int s = 0; s < kBigSteps; ++s) {
      const f32x4 dz = bc0 + 4 * s + q < b_hi ? dy[s] : zero;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        colsum[i] += dz[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(dz[i], x[s][t], acc[i][t]);
      }
    }
  };
  if (b_lo < b_hi) {
    int bc0 = b_lo;
    load(dyA, xA, bc0);
    while (true) {
      const int nb = bc0 + 4 * kBigSteps;
      if (nb < b_hi) load(dyB, xB, nb);
      mul(dyA, xA, bc0);
      if (nb >= b_hi) break;
      const int nb2 = nb + 4 * kBigSteps;
      if (nb2 < b_hi) load(dyA, xA, nb2);
      mul(dyB, xB, nb);
      if (nb2 >= b_hi) break;
      bc0 = nb2;
    }
  }
  // bias gradient: the four row groups' serial sums, added pairwise (q0 + q1) + (q2 + q3)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    colsum[i] += __shfl_xor(colsum[i], 16, 64);
    colsum[i] += __shfl_xor(colsum[i], 32, 64);
  }
  const size_t wk = (size_t)a.N * a.K;
  float* __restrict__ dW = a.dW[item];
  float* __restrict__ db = a.db[item];
  float* part = a.slices > 1 ? a.partial + ((size_t)item * a.slices + slice) * (wk + a.N) : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const size_t o = (size_t)(n0 + 4 * (4 * q + g) + i) * a.K + k0 + 4 * r;
      f32x4 v = {acc[i][0][g], acc[i][1][g], acc[i][2][g], acc[i][3][g]};
      if (part) *reinterpret_cast<f32x4*>(part + o) = v;
      else {
        if (a.accumulate) {
          const f32x4 old = *reinterpret_cast<const f32x4*>(dW + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += old[e];
        }
        *reinterpret_cast<f32x4*>(dW + o) = v;
      }
    }
  if (kg == 0 && q == 0) {
    if (part) *reinterpret_cast<f32x4*>(part + wk + n0 + 4 * r) = colsum;
    else if (db) {
#pragma unroll
      for (int e = 0; e < 4; ++e) db[n0 + 4 * r + e] = a.accumulate ? db[n0 + 4 * r + e] + colsum[e] : colsum[e];
    }
  }
}

// N and K multiples of 128: the four 64 x 64 blocks of a 128 x 128 block of dW are ONE workgroup's four waves and read the
// same rows — as four independent streams (above) every operand row crosses the CU's memory pipe twice, 2 KB per row at
// N = K = 128, and the pipe is the bound (122 us against 57 of f32 MFMA at 262144 rows).  Here the workgroup loads each
// 16-row batch of dY and X ONCE, coalesced (a thread's four float4), into a double-buffered LDS stage and the waves read
// their operands from there (rows padded to 132 floats: the four row groups of a wave's 16-byte reads start 16 bytes
// apart); one barrier per batch, the next batch's global loads in flight under the current batch's 64 MFMAs per wave.
// Every wave issues the MFMAs of the kernel above on the same values in the same order: the same bits.
constexpr int kStagePitch = 132;
__global__ __launch_bounds__(256) void lin_bwd_weight_big_lds_kernel(const LinBwdW a) {
  __shared__ __attribute__((aligned(16))) float stage[2][2][16 * kStagePitch];   // [buffer][dY | X][16 rows][128 + pad]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4, wave = tid >> 6;
  const int item = blockIdx.y, slice = blockIdx.z;
  const int kb128 = a.K >> 7;
  const int nb = blockIdx.x / kb128, kb = blockIdx.x - nb * kb128;
  const int ng = 2 * nb + (wave >> 1), kg = 2 * kb + (wave & 1);        // this wave's 64 x 64 block
  const int n0 = ng * 64, k0 = kg * 64;
  const int b_lo = slice * a.rows_per_slice;
  const int b_hi = min(a.B, b_lo + a.rows_per_slice);
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = zero;
  f32x4 colsum = zero;
  // the workgroup's loads of a batch: float4 number j * 256 + tid of the [16 rows][32 float4] tile of each matrix
  const float* __restrict__ dYg = a.dY[item] + 128 * nb;
  const float* __restrict__ Xg = a.X[item] + 128 * kb;
  f32x4 ld[2][2];
  auto fetch = [&](int bc0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f4 = j * 256 + tid, row = f4 >> 5, c4 = f4 & 31;
      const int bb = bc0 + row;
      const size_t bs = (size_t)(bb < b_hi ? bb : b_lo);
      ld[0][j] = *reinterpret_cast<const f32x4*>(dYg + bs * a.ldy + 4 * c4);
      ld[1][j] = *reinterpret_cast<const f32x4*>(Xg + bs * a.ldx + 4 * c4);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f4 = j * 256 + tid, row = f4 >> 5, c4 = f4 & 31;
      *reinterpret_cast<f32x4*>(&stage[buf][0][row * kStagePitch + 4 * c4]) = ld[0][j];
      *reinterpret_cast<f32x4*>(&stage[buf][1][row * kStagePitch + 4 * c4]) = ld[1][j];
    }
  };
  if (b_lo < b_hi) {
    fetch(b_lo);
    park(0);
    __syncthreads();
    int buf = 0;
    for (int bc0 = b_lo; bc0 < b_hi; bc0 += 4 * kBigSteps, buf ^= 1) {
      const bool more = bc0 + 4 * kBigSteps < b_hi;
      if (more) fetch(bc0 + 4 * kBigSteps);
      const float* sy = &stage[buf][0][64 * (wave >> 1) + 4 * r];
      const float* sx = &stage[buf][1][64 * (wave & 1) + 4 * r];
#pragma unroll
      for (int s2 = 0; s2 < kBigSteps; ++s2) {
        const int row = 4 * s2 + q;
        const f32x4 dy = *reinterpret_cast<const f32x4*>(sy + row * kStagePitch);
        const f32x4 x = *reinterpret_cast<const f32x4*>(sx + row * kStagePitch);
        const f32x4 dz = bc0 + row < b_hi ? dy : zero;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          colsum[i] += dz[i];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(dz[i], x[t], acc[i][t]);
        }
      }
      if (more) park(buf ^ 1);
      __syncthreads();
    }
  }
  // bias gradient: the four row groups' serial sums, added pairwise (q0 + q1) + (q2 + q3)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    colsum[i] += __shfl_xor(colsum[i], 16, 64);
    colsum[i] += __shfl_xor(colsum[i], 32, 64);
  }
  const size_t wk = (size_t)a.N * a.K;
  float* __restrict__ dW = a.dW[item];
  float* __restrict__ db = a.db[item];
  float* part = a.slices > 1 ? a.partial + ((size_t)item * a.slices + slice) * (wk + a.N) : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const size_t o = (size_t)(n0 + 4 * (4 * q + g) + i) * a.K + k0 + 4 * r;
      f32x4 v = {acc[i][0][g], acc[i][1][g], acc[i][2][g], acc[i][3][g]};
      if (part) *reinterpret_cast<f32x4*>(part + o) = v;
      else {
        if (a.accumulate) {
          const f32x4 old = *reinterpret_cast<const f32x4*>(dW + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += old[e];
        }
        *reinterpret_cast<f32x4*>(dW + o) = v;
      }
    }
  if (kg == 0 && q == 0) {
    if (part) *reinterpret_cast<f32x4*>(part + wk + n0 + 4 * r) = colsum;
    else if (db) {
#pragma unroll
      for (int e = 0; e < 4; ++e) db[n0 + 4 * r + e] = a.accumulate ? db[n0 + 4 * r + e] + colsum[e] : colsum[e];
    }
  }
}

// dW / db = the slices' partials added in a fixed order: eight groups of consecutive slices, each ascending, then the groups ascending
__global__ __launch_bounds__(256) void lin_slice_reduce_kernel(const LinBwdW a) {
  __shared__ float grp[8][32];
  const int item = blockIdx.y, col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const size_t wk = (size_t)a.N * a.K, per = wk + a.N;
  const size_t i = (size_t)blockIdx.x * 32 + col;
  float s = 0.0f;
  if (i < per) {
    const int each = (a.slices + 7) / 8;
    const int k0 = sl * each, k1 = min(a.slices, k0 + each);
    const float* p = a.partial + (size_t)item * a.slices * per + i;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) s += p[(size_t)k * per];
  }
  grp[sl][col] = s;
  __syncthreads();
  if (sl != 0 || i >= per) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) s += grp[k][col];
  float* dst = i < wk ? a.dW[item] + i : (a.db[item] ? a.db[item] + (i - wk) : nullptr);
  if (dst) *dst = a.accumulate ? *dst + s : s;
}

// ---- NoisyLinear heads (rainbow_dqn_cartpole.py:60-95): effective weights and the gradients' way back -------------
constexpr int kNoisyLayers = GYMRL_NOISY_MAX_LAYERS;
struct NoisyArgs {
  const float* w_mu[kNoisyLayers]; const float* w_sigma[kNoisyLayers]; const float* w_eps[kNoisyLayers];
  const float* b_mu[kNoisyLayers]; const float* b_sigma[kNoisyLayers]; const float* b_eps[kNoisyLayers];
  float* w_eps_copy[kNoisyLayers]; float* b_eps_copy[kNoisyLayers];
  uint64_t seed[kNoisyLayers], counter[kNoisyLayers]; const uint64_t* counter_dev[kNoisyLayers];
  int draw[kNoisyLayers];           // combine: generate this layer's noise here (w_eps / b_eps are not read)
  int eval[kNoisyLayers];           // this layer contributes its mu only (a target network beside training-mode layers)
  float* dw_mu[kNoisyLayers]; float* dw_sigma[kNoisyLayers]; float* db_mu[kNoisyLayers]; float* db_sigma[kNoisyLayers];
  int row0[kNoisyLayers + 1];       // first stacked row of each layer
  int n_layers, K, training, accumulate;
  float* W; float* b;               // stacked [rows, K], [rows]
  const float* dW; const float* db;
  // gymrl_noisy_combine_images: workgroups combine_blocks .. gridDim.x - 1 rebuild weight images instead
  int combine_blocks, n_img;
  const float* img_W[GYMRL_NOISY_MAX_IMAGES]; float* img_f[GYMRL_NOISY_MAX_IMAGES]; float* img_b[GYMRL_NOISY_MAX_IMAGES];
  int img_H[GYMRL_NOISY_MAX_IMAGES];
};

// lin_device.hpp's img_fwd_index / img_bwd_index by DESTINATION float4 j = (tile * steps + c) * 64 + 16 q + r: coalesced stores
__device__ __forceinline__ void pack_images(const NoisyArgs& a, int pb, int npb) {
  for (int i = 0; i < a.n_img; ++i) {
    const int H = a.img_H[i], steps = H >> 4;
    const float* __restrict__ W = a.img_W[i];
    const int64_t nv = (int64_t)H * H / 4;
    for (int64_t j = (int64_t)pb * 256 + threadIdx.x; j < nv; j += (int64_t)npb * 256) {
      const int lane = (int)(j & 63), c = (int)((j >> 6) % steps), tile = (int)((j >> 6) / steps), r = lane & 15, q = lane >> 4;
      if (a.img_f[i])
        reinterpret_cast<f32x4*>(a.img_f[i])[j] = *reinterpret_cast<const f32x4*>(W + (size_t)(16 * tile + r) * H + 16 * c + 4 * q);
      if (a.img_b[i]) {
        const float* w = W + (size_t)(16 * c + 4 * q) * H + 16 * tile + r;
        const f32x4 v = {w[0], w[H], w[2 * (size_t)H], w[3 * (size_t)H]};
        reinterpret_cast<f32x4*>(a.img_b[i])[j] = v;
      }
    }
  }
}

// W[row0 + n] = mu + sigma * eps (training) | mu (eval); the noise is also copied through to the module's buffers
__global__ __launch_bounds__(256) void noisy_combine_kernel(const NoisyArgs a) {
  const int rows = a.row0[a.n_layers];
  const int64_t total = (int64_t)rows * (a.K + 1);
  const int cb = a.n_img > 0 ? a.combine_blocks : (int)gridDim.x;
  if ((int)blockIdx.x >= cb) { pack_images(a, (int)blockIdx.x - cb, (int)gridDim.x - cb); return; }
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)cb * 256) {
    const int row = (int)(t / (a.K + 1)), k = (int)(t % (a.K + 1));
    int l = 0;
    while (l + 1 < a.n_layers && row >= a.row0[l + 1]) ++l;
    const int n = row - a.row0[l];
    // draw: NoisyLinear.reset_noise() of this forward happens here — the very values gymrl_noisy_noise writes
    float fj = 0.0f;
    uint64_t ctr = 0;
    const bool training = a.training && !a.eval[l];
    if (training && a.draw[l]) {
      ctr = a.counter_dev[l] ? a.counter_dev[l][0] : a.counter[l];
      fj = scale_noise(box_muller(a.seed[l], ctr, 1u, (uint32_t)n));
    }
    if (k < a.K) {
      const size_t o = (size_t)n * a.K + k;
      float w = a.w_mu[l][o];
      if (training) {
        const float e = a.draw[l] ? fj * scale_noise(box_muller(a.seed[l], ctr, 0u, (uint32_t)k)) : a.w_eps[l][o];
        w = w + a.w_sigma[l][o] * e;
        if (a.w_eps_copy[l]) a.w_eps_copy[l][o] = e;
      }
      a.W[(size_t)row * a.K + k] = w;
    } else {
      float bv = a.b_mu[l][n];
      if (training) {
        const float e = a.draw[l] ? fj : a.b_eps[l][n];
        bv = bv + a.b_sigma[l][n] * e;
        if (a.b_eps_copy[l]) a.b_eps_copy[l][n] = e;
      }
      a.b[row] = bv;
    }
  }
}

// d mu = dW, d sigma = dW * eps (and the biases alike), straight into the parameters' gradient views
__global__ __launch_bounds__(256) void noisy_split_kernel(const NoisyArgs a) {
  const int rows = a.row0[a.n_layers];
  const int64_t total = (int64_t)rows * (a.K + 1);
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int row = (int)(t / (a.K + 1)), k = (int)(t % (a.K + 1));
    int l = 0;
    while (l + 1 < a.n_layers && row >= a.row0[l + 1]) ++l;
    const int n = row - a.row0[l];
    if (k < a.K) {
      const size_t o = (size_t)n * a.K + k;
      const float g = a.dW[(size_t)row * a.K + k];
      a.dw_mu[l][o] = a.accumulate ? a.dw_mu[l][o] + g : g;
      if (a.dw_sigma[l]) {
        const float gs = a.training ? g * a.w_eps[l][o] : 0.0f;
        a.dw_sigma[l][o] = a.accumulate ? a.dw_sigma[l][o] + gs : gs;
      }
    } else {
      const float g = a.db[row];
      a.db_mu[l][n] = a.accumulate ? a.db_mu[l][n] + g : g;
      if (a.db_sigma[l]) {
        const float gs = a.training ? g * a.b_eps[l][n] : 0.0f;
        a.db_sigma[l][n] = a.accumulate ? a.db_sigma[l][n] + gs : gs;
      }
    }
  }
}

// backward of q = v + a - mean(a): dS[:, j < A] = dq_j - mean(dq), dS[:, A] = sum(dq)
__global__ __launch_bounds__(256) void dueling_bwd_kernel(const float* __restrict__ dq, int B, int A, float* __restrict__ dS) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float s = 0.0f;
  for (int j = 0; j < A; ++j) s += dq[(size_t)b * A + j];
  const float m = s / (float)A;
  for (int j = 0; j < A; ++j) dS[(size_t)b * (A + 1) + j] = dq[(size_t)b * A + j] - m;
  dS[(size_t)b * (A + 1) + A] = s;
}

// ---- skinny shapes at large B: a 16 x 16 MFMA tile per wave is store-issue bound there (dword stores of 64-byte row
// segments); with <= 16 terms per output the products fit the VALU and a lane writes 16 bytes ----------------------
constexpr int kSkinny = 16;

// dX[b][k .. k+3] = sum_n (dY[b][n] * act'(Y[b][n])) W[n][k .. k+3],  N <= 16
__global__ __launch_bounds__(256) void lin_bwd_input_skinny_kernel(const LinBwdIn a) {
  const int item = blockIdx.y;
  const float* __restrict__ dY = a.dY[item];
  const float* __restrict__ Yv = a.Y[item];
  const float* __restrict__ W = a.W[item];
  float* __restrict__ dX = a.dX[item];
  const int act = a.act[item];
  const float lo = a.lo[item], hi = a.hi[item];
  const int k4 = a.K >> 2;
  const int64_t total = (int64_t)a.B * k4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t b = t / k4;
    const int k = (int)(t % k4) * 4;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int n = 0; n < a.N; ++n) {
      float dz = dY[b * a.ldy + n];
      if (act != GYMRL_ACT_NONE) dz *= act_bwd(Yv[b * a.ldy + n], act, lo, hi);
      const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)n * a.K + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += dz * w[e];
    }
    f32x4* dst = reinterpret_cast<f32x4*>(dX + b * a.lddx + k);
    if (a.accumulate) { const f32x4 o = *dst; s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3]; }
    *dst = s;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// 16 x 64 tiles per wave once 16 x 16 tiles would be more waves than the chip has SIMDs twice over
inline int pick_nt(int row_tiles, int cols) { return (cols > 16 && (int64_t)row_tiles * cdiv(cols, 16) > 2048) ? 4 : 1; }
// row slices of the weight gradient: enough (tile, slice) waves to fill the chip — a skinny layer has few output tiles, so
// at 262144 rows its reduction is cut 128 ways (16 slices left 256 waves walking 16384 rows each: 440 us) — and never
// fewer than 256 rows per slice
inline bool big_shape(int B, int N, int K) { return lin::bwd_weight_big_shape(B, N, K); }     // (lin_device.hpp: shared with the
inline int slices_for(int B, int N, int K) { return lin::bwd_weight_slices(B, N, K); }           //  fused step's weight-gradient tiles)

}  // namespace

extern "C" {

size_t gymrl_lin_workspace_bytes(int B, int N, int K, int n_items) {
  const int s = slices_for(B, N, K);
  return s > 1 ? sizeof(float) * (size_t)n_items * s * ((size_t)N * K + N) : 0;
}

int gymrl_lin_fwd(const gymrl_lin_item* items, int n_items, int B, int K, int K1, int N, int ldx, int ldx2, int ldy,
                  void* stream_) {
  if (!items || n_items < 1 || n_items > kItems || B < 0 || K < 1 || N < 1 || K1 < 0 || K1 > K) return -22;
  if (B == 0) return 0;
  LinFwd a{};
  bool vec = K1 == K && K % 4 == 0 && ldx % 4 == 0;
  for (int i = 0; i < kItems; ++i) {
    const gymrl_lin_item& it = items[i < n_items ? i : 0];
    if (!it.x || !it.w || !it.y || (K1 < K && !it.x2) || it.act < GYMRL_ACT_NONE || it.act > GYMRL_ACT_SILU ||
        (it.act == GYMRL_ACT_DUELING && (N < 2 || N > 16)))
      return -22;
    a.act[i] = it.act; a.lo[i] = it.lo; a.hi[i] = it.hi;
    a.X[i] = it.x; a.X2[i] = K1 < K ? it.x2 : nullptr; a.W[i] = it.w; a.b[i] = it.b; a.Y[i] = it.y;
    a.argmax[i] = it.act == GYMRL_ACT_DUELING ? it.argmax : nullptr;
    vec = vec && aligned16(it.x) && aligned16(it.w);
  }
  a.B = B; a.K = K; a.K1 = K1; a.N = N; a.ldx = ldx; a.ldx2 = ldx2; a.ldy = ldy;
  const int row_tiles = cdiv(B, 16), nt = pick_nt(row_tiles, N);
  a.col_groups = cdiv(N, 16 * nt);
  const dim3 grid(cdiv(row_tiles * a.col_groups, kWavesPerBlock), n_items), block(64 * kWavesPerBlock);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (nt == 4) {
    if (vec) hipLaunchKernelGGL((lin_fwd_kernel<4, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((lin_fwd_kernel<4, false>), grid, block, 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((lin_fwd_kernel<1, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((lin_fwd_kernel<1, false>), grid, block, 0, s, a);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_lin_bwd_input(const gymrl_lin_item* items, int n_items, int B, int N, int K, int K1, int ldy, int lddx,
                        int lddx2, int accumulate, int sum_items, void* stream_) {
  if (!items || n_items < 1 || n_items > kItems || B < 0 || K < 1 || N < 1 || K1 < 0 || K1 > K) return -22;
  if (B == 0) return 0;
  LinBwdIn a{};
  bool vec = N % 4 == 0 && ldy % 4 == 0;
  for (int i = 0; i < kItems; ++i) {
    const gymrl_lin_item& it = items[i < n_items ? i : 0];
    if (!it.dy || !it.w || it.act < GYMRL_ACT_NONE || it.act > GYMRL_ACT_CLAMP || (it.act != GYMRL_ACT_NONE && !it.y) ||
        (!it.dx && !it.dx2))
      return -22;
    a.act[i] = it.act; a.lo[i] = it.lo; a.hi[i] = it.hi;
    a.dY[i] = it.dy; a.Y[i] = it.y; a.W[i] = it.w; a.dX[i] = it.dx; a.dX2[i] = it.dx2;
    vec = vec && aligned16(it.dy) && (it.act == GYMRL_ACT_NONE || aligned16(it.y));
  }
  a.B = B; a.N = N; a.K = K; a.K1 = K1; a.ldy = ldy; a.lddx = lddx; a.lddx2 = lddx2;
  a.accumulate = accumulate;
  a.n_sum = sum_items ? n_items : 1;
  const int row_tiles = cdiv(B, 16), nt = pick_nt(row_tiles, K);
  a.col_groups = cdiv(K, 16 * nt);
  const dim3 grid(cdiv(row_tiles * a.col_groups, kWavesPerBlock), sum_items ? 1 : n_items), block(64 * kWavesPerBlock);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  bool skinny = N <= kSkinny && K1 == K && K % 4 == 0 && lddx % 4 == 0 && B > 8192 && !(sum_items && n_items > 1);
  for (int i = 0; i < n_items; ++i) skinny = skinny && items[i].dx && aligned16(items[i].dx) && aligned16(items[i].w);
  if (skinny) {
    int64_t nb = ((int64_t)B * (K / 4) + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(lin_bwd_input_skinny_kernel, dim3((unsigned)nb, n_items), dim3(256), 0, s, a);
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  if (nt == 4) {
    if (vec) hipLaunchKernelGGL((lin_bwd_input_kernel<4, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((lin_bwd_input_kernel<4, false>), grid, block, 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((lin_bwd_input_kernel<1, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((lin_bwd_input_kernel<1, false>), grid, block, 0, s, a);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_lin_bwd_weight(const gymrl_lin_item* items, int n_items, int B, int N, int K, int K1, int ldy, int ldx,
                         int ldx2, int accumulate, void* workspace, void* stream_) {
  if (!items || n_items < 1 || n_items > kItems || B < 0 || K < 1 || N < 1 || K1 < 0 || K1 > K) return -22;
  LinBwdW a{};
  for (int i = 0; i < kItems; ++i) {
    const gymrl_lin_item& it = items[i < n_items ? i : 0];
    if (!it.dy || !it.x || !it.dw || it.act < GYMRL_ACT_NONE || it.act > GYMRL_ACT_CLAMP ||
        (it.act != GYMRL_ACT_NONE && !it.y) || (K1 < K && !it.x2))
      return -22;
    a.act[i] = it.act; a.lo[i] = it.lo; a.hi[i] = it.hi;
    a.dY[i] = it.dy; a.Y[i] = it.y; a.X[i] = it.x; a.X2[i] = K1 < K ? it.x2 : nullptr; a.dW[i] = it.dw; a.db[i] = it.db;
  }
  a.B = B; a.N = N; a.K = K; a.K1 = K1; a.ldy = ldy; a.ldx = ldx; a.ldx2 = ldx2;
  a.accumulate = accumulate;
  a.slices = slices_for(B, N, K);
  a.rows_per_slice = lin::bwd_weight_rows_per_slice(B, a.slices);
  if (a.slices > 1 && !workspace) return -22;
  a.partial = static_cast<float*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  bool big = big_shape(B, N, K) && K1 == K && ldy % 4 == 0 && ldx % 4 == 0;
  for (int i = 0; i < n_items; ++i)
    big = big && items[i].act == GYMRL_ACT_NONE && aligned16(items[i].dy) && aligned16(items[i].x) && aligned16(items[i].dw);
  if (big) {
    const dim3 grid(cdiv((N / 64) * (K / 64), kWavesPerBlock), n_items, a.slices), block(64 * kWavesPerBlock);
    if (N % 128 == 0 && K % 128 == 0)
      hipLaunchKernelGGL(lin_bwd_weight_big_lds_kernel, dim3((N / 128) * (K / 128), n_items, a.slices), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(lin_bwd_weight_big_kernel, grid, block, 0, s, a);
    if (a.slices > 1) {
      const size_t per = (size_t)N * K + N;
      hipLaunchKernelGGL(lin_slice_reduce_kernel, dim3((unsigned)((per + 31) / 32), n_items), dim3(256), 0, s, a);
    }
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  const int n_tiles = cdiv(N, 16);
  const int nt = (int64_t)n_tiles * cdiv(K, 16) * a.slices > 4096 ? 4 : 1;
  a.col_groups = cdiv(K, 16 * nt);
  const dim3 grid(cdiv(n_tiles * a.col_groups, kWavesPerBlock), n_items, a.slices), block(64 * kWavesPerBlock);
  if (nt == 4) hipLaunchKernelGGL((lin_bwd_weight_kernel<4>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((lin_bwd_weight_kernel<1>), grid, block, 0, s, a);
  if (a.slices > 1) {
    const size_t per = (size_t)N * K + N;
    hipLaunchKernelGGL(lin_slice_reduce_kernel, dim3((unsigned)((per + 31) / 32), n_items), dim3(256), 0, s, a);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

static int noisy_fill(NoisyArgs& a, const gymrl_noisy_layer* layers, int n_layers, int K, int training) {
  if (!layers || n_layers < 1 || n_layers > kNoisyLayers || K < 1) return -22;
  int rows = 0;
  for (int l = 0; l < n_layers; ++l) {
    const gymrl_noisy_layer& L = layers[l];
    const bool tr = training && !L.eval;
    if (!L.w_mu || !L.b_mu || L.n_out < 1 || (tr && (!L.w_sigma || !L.b_sigma))) return -22;
    if (tr && !L.draw && (!L.w_eps || !L.b_eps)) return -22;
    if ((L.w_eps_copy == nullptr) != (L.b_eps_copy == nullptr)) return -22;
    a.seed[l] = L.seed; a.counter[l] = L.counter; a.counter_dev[l] = L.counter_dev; a.draw[l] = L.draw; a.eval[l] = L.eval;
    a.w_mu[l] = L.w_mu; a.w_sigma[l] = L.w_sigma; a.w_eps[l] = L.w_eps;
    a.b_mu[l] = L.b_mu; a.b_sigma[l] = L.b_sigma; a.b_eps[l] = L.b_eps;
    a.w_eps_copy[l] = L.w_eps_copy; a.b_eps_copy[l] = L.b_eps_copy;
    a.dw_mu[l] = L.dw_mu; a.dw_sigma[l] = L.dw_sigma; a.db_mu[l] = L.db_mu; a.db_sigma[l] = L.db_sigma;
    a.row0[l] = rows;
    rows += L.n_out;
  }
  a.row0[n_layers] = rows;
  a.n_layers = n_layers; a.K = K; a.training = training;
  return rows;
}

int gymrl_noisy_combine(const gymrl_noisy_layer* layers, int n_layers, int K, int training, float* W_out, float* b_out,
                        void* stream_) {
  NoisyArgs a{};
  const int rows = noisy_fill(a, layers, n_layers, K, training);
  if (rows < 0 || !W_out || !b_out) return -22;
  a.W = W_out; a.b = b_out;
  const int64_t total = (int64_t)rows * (K + 1);
  hipLaunchKernelGGL(noisy_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_noisy_combine_images(const gymrl_noisy_layer* layers, int n_layers, int K, int training, float* W_out, float* b_out,
                               const gymrl_weight_image* images, int n_images, void* stream_) {
  NoisyArgs a{};
  const int rows = noisy_fill(a, layers, n_layers, K, training);
  if (rows < 0 || !W_out || !b_out || n_images < 0 || n_images > GYMRL_NOISY_MAX_IMAGES || (n_images > 0 && !images)) return -22;
  a.W = W_out; a.b = b_out;
  int pack_blocks = 0;
  for (int i = 0; i < n_images; ++i) {
    const gymrl_weight_image& im = images[i];
    if (!im.W || im.H < 16 || (im.H & 15) != 0 || (!im.img_fwd && !im.img_bwd)) return -22;
    a.img_W[i] = im.W; a.img_H[i] = im.H; a.img_f[i] = im.img_fwd; a.img_b[i] = im.img_bwd;
    pack_blocks += (im.H * im.H / 4 + 255) / 256;
  }
  const int64_t total = (int64_t)rows * (K + 1);
  a.combine_blocks = (int)((total + 255) / 256);
  a.n_img = n_images;
  hipLaunchKernelGGL(noisy_combine_kernel, dim3((unsigned)(a.combine_blocks + pack_blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_noisy_split(const gymrl_noisy_layer* layers, int n_layers, int K, int training, const float* dW, const float* db,
                      int accumulate, void* stream_) {
  NoisyArgs a{};
  const int rows = noisy_fill(a, layers, n_layers, K, training);
  if (rows < 0 || !dW || !db) return -22;
  for (int l = 0; l < n_layers; ++l)
    if (!layers[l].dw_mu || !layers[l].db_mu || (!layers[l].dw_sigma) != (!layers[l].db_sigma)) return -22;
  a.dW = dW; a.db = db; a.accumulate = accumulate;
  const int64_t total = (int64_t)rows * (K + 1);
  hipLaunchKernelGGL(noisy_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), a);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_dueling_bwd(const float* dq, int B, int A, float* dS_out, void* stream_) {
  if (!dq || !dS_out || B < 0 || A < 1) return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(dueling_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), dq, B, A,
                     dS_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
