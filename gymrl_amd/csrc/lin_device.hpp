// lin_device.hpp — the 16 x 16 MFMA tile bodies of the off-policy networks' layers as device functions.
//
// lin.hip runs every Linear(+activation) of the DQN / Rainbow / SAC / TD3 / DDPG networks as one launch per layer and
// direction: a wavefront owns a 16 x 16 output tile and walks the whole reduction with v_mfma_f32_16x16x4_f32.  At the
// reference's batch sizes (128 / 256 rows) such a launch is ~1 us of work behind ~4-7 us of dispatch, first-load latency
// and drain, and a SAC update is ~50 of them.  The fused step kernels (offpolicy_step.hip) keep a 16-row slab of the
// batch in ONE workgroup's LDS through a whole chain of layers — rows never interact in a forward or input-gradient
// pass — so a layer costs its tile's MFMA chain and a workgroup barrier instead of a launch.  The tile bodies below are
// what both use: the SAME sequence of MFMAs on the SAME operands in the SAME order (documented in lin.hip's header), so
// the fused path is bit-identical to the layer-by-layer path and every parity test of the latter pins the former.
//
//   tile_fwd        acc = X[16 x K] . W[n-tile]^T          X from LDS (row stride ldx, optional second block X2)
//   tile_bwd_input  acc = dZ[16 x N] . W[:, k-tile]        dZ from LDS
//   tile_bwd_weight acc = dZ[B x n-tile]^T . X[B x k-tile] dZ, X from global (the weight gradient reduces over the batch)
//
// Lane (r = lane & 15, q = lane >> 4) of the result holds acc[g] = out[row 4q + g][col r] of the tile (rows of the slab /
// of the n-tile for tile_bwd_weight).
#pragma once
#include "train_device.hpp"
#include "../../include/gymrl.h"

namespace gymrl {
namespace lin {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float act_fwd(float z, int act, float lo, float hi) {
  if (act == GYMRL_ACT_RELU) return fmaxf(z, 0.0f);
  if (act == GYMRL_ACT_TANH) return train_tanhf(z);
  if (act == GYMRL_ACT_CLAMP) return fminf(fmaxf(z, lo), hi);
  if (act == GYMRL_ACT_SILU) return z / (1.0f + expf(-z));          // forward only (its derivative needs z, not y)
  return z;
}
// d act / d z as a function of the saved OUTPUT y
__device__ __forceinline__ float act_bwd(float y, int act, float lo, float hi) {
  if (act == GYMRL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (act == GYMRL_ACT_TANH) return 1.0f - y * y;
  if (act == GYMRL_ACT_CLAMP) return (y > lo && y < hi) ? 1.0f : 0.0f;
  return 1.0f;
}

constexpr int kSlab = 16;                 // rows of a slab = rows of an MFMA tile

// Row stride (floats) of a [16][K] slab in LDS: K rounded up to a multiple of 4, plus 4 — 16-byte rows for the f32x4
// reads, and the 16 rows of a quarter-wave land in 16 different 16-byte bank groups when K is a multiple of 64.
__host__ __device__ __forceinline__ int slab_ld(int K) { return ((K + 3) & ~3) + 4; }

// The reduction of a slab tile is at most kMaxSteps 16-deep MFMA steps per pass (K <= 256: the fused kernels' limit; longer
// reductions take further passes).  ALL of a pass's weight loads are issued before its first MFMA — a slab stage is one
// L2 round trip plus the MFMA chain, not a round trip per step (the first version, with the loads inside the loop, ran a
// 256 x 256 layer in ~6 us per stage; the MFMA chain is 0.85 us) — and the X operand comes from LDS step by step.
constexpr int kMaxSteps = 16;

// tile_fwd's vector branch (K1 == K, K % 4 == 0).  KC: the reduction length as a compile-time constant (0: take Kr — itself a
// constant after inlining in the kernels built for one hidden width): with K = 256 known, the sixteen "beyond K?" tests, their
// selects and the loop's exits fold away, which in a narrow layer's lone chain of 64 MFMAs was most of the instructions between them.
template <int KC>
__device__ __forceinline__ f32x4 tile_fwd_vec(const float* xrow, const float* wrow, int Kr, bool n_ok, int q) {
  const int K = KC ? KC : Kr;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  {
    for (int kp = 0; kp < K; kp += 16 * kMaxSteps) {
      f32x4 wb[kMaxSteps];
#pragma unroll
      for (int c = 0; c < kMaxSteps; ++c) {
        if (kp + 16 * c >= K) break;                          // (wave-uniform)
        const int k = kp + 16 * c + 4 * q;
        wb[c] = *reinterpret_cast<const f32x4*>(wrow + (k < K ? k : 0));
      }
      // the X operand of step c + 2 is requested before step c's four MFMAs (a lone chain — a narrow layer is ONE tile on one
      // wave — otherwise waits out an LDS round trip per step: 16 x ~120 clocks on top of the chain's 64 x 32)
      auto xread = [&](int c) {
        const int k = kp + 16 * c + 4 * q;
        const bool k_ok = k < K;
        const f32x4 xa = *reinterpret_cast<const f32x4*>(xrow + (k_ok ? k : 0));
        return k_ok ? xa : zero;
      };
      f32x4 xq[2] = {xread(0), kp + 16 < K ? xread(1) : zero};
#pragma unroll
      for (int c = 0; c < kMaxSteps; ++c) {
        if (kp + 16 * c >= K) break;
        const bool k_ok = kp + 16 * c + 4 * q < K;
        const f32x4 x = xq[c & 1];
        if (c + 2 < kMaxSteps && kp + 16 * (c + 2) < K) xq[c & 1] = xread(c + 2);
        const f32x4 w = (n_ok && k_ok) ? wb[c] : zero;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma16(x[e], w[e], acc);
        __builtin_amdgcn_sched_barrier(0);                    // (the reads stay two steps ahead: hoisted to the top they spill)
      }
    }
  }
  return acc;
}

// acc = X . W[nb .. nb + 15]^T for the slab X = [Xs | X2s] ([16][K1] and [16][K - K1] in LDS; K1 == K: one block).
// The order of lin_fwd_kernel: k0 = 0, 16, ...; e = 0..3; the MFMA adds k = k0 + 4q + e over q.  Rows beyond the slab's
// valid rows must hold zeros (the callers zero-fill), columns n >= N and k >= K contribute exact zeros.
__device__ __forceinline__ f32x4 tile_fwd(const float* Xs, int ldx, const float* X2s, int ldx2, int K, int K1,
                                          const float* __restrict__ W, int N, int nb, int lane) {
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const int n = nb + r;
  const bool n_ok = n < N;
  const float* wrow = W + (size_t)(n_ok ? n : 0) * K;
  const float* xrow = Xs + r * ldx;
  if (K1 == K && (K & 3) == 0) {
    acc = tile_fwd_vec<0>(xrow, wrow, K, n_ok, q);           // (K is a constant after inlining where the kernel is built for one H)
  } else {
    const float* x2row = X2s ? X2s + r * ldx2 : nullptr;
    for (int k0 = 0; k0 < K; k0 += 16) {
      const int k = k0 + 4 * q;
      float xa[4], wb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kk = k + e;
        float v = 0.0f;
        if (kk < K) v = kk < K1 ? xrow[kk] : x2row[kk - K1];
        xa[e] = v;
        wb[e] = (n_ok && kk < K) ? W[(size_t)n * K + kk] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma16(xa[e], wb[e], acc);
    }
  }
  return acc;
}

// tile_fwd's vector branch for TWO slabs that share the weight fragment (acting on more envs than the chip has compute units x 16:
// one pass of 32-row workgroups instead of two of 16-row ones — the weights are streamed once and feed two MFMA chains).
// K1 == K, K % 4 == 0.  Each slab's accumulator sees exactly tile_fwd's sequence.
__device__ __forceinline__ void tile_fwd_x2(const float* Xs0, const float* Xs1, int ldx, int K, const float* __restrict__ W, int N, int nb,
                                            int lane, f32x4& acc0, f32x4& acc1) {
  const int r = lane & 15, q = lane >> 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  acc0 = zero; acc1 = zero;
  const int n = nb + r;
  const bool n_ok = n < N;
  const float* wrow = W + (size_t)(n_ok ? n : 0) * K;
  const float* x0 = Xs0 + r * ldx;
  const float* x1 = Xs1 + r * ldx;
  for (int kp = 0; kp < K; kp += 16 * kMaxSteps) {
    f32x4 wb[kMaxSteps];
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (kp + 16 * c >= K) break;
      const int k = kp + 16 * c + 4 * q;
      wb[c] = *reinterpret_cast<const f32x4*>(wrow + (k < K ? k : 0));
    }
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (kp + 16 * c >= K) break;
      const int k = kp + 16 * c + 4 * q;
      const bool k_ok = k < K;
      const f32x4 xa = *reinterpret_cast<const f32x4*>(x0 + (k_ok ? k : 0));
      const f32x4 xb = *reinterpret_cast<const f32x4*>(x1 + (k_ok ? k : 0));
      const f32x4 a0 = k_ok ? xa : zero, a1 = k_ok ? xb : zero;
      const f32x4 w = (n_ok && k_ok) ? wb[c] : zero;
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc0 = mfma16(a0[e], w[e], acc0); acc1 = mfma16(a1[e], w[e], acc1); }
    }
  }
}

// tile_bwd_input(tile_bwd_input(acc, dZa, Wa), dZb, Wb) — the summed input gradient of an input two layers share, ONE
// accumulator chain (layer a's terms, then layer b's) — with BOTH layers' weights requested before the first MFMA: as two
// calls the second layer's 64 scalar loads wait behind the first layer's dependent chain.  N <= 16 * kMaxSteps, N % 4 == 0.
__device__ __forceinline__ f32x4 tile_bwd_input_pair(f32x4 acc, const float* dZa, const float* dZb, int ldz, int N,
                                                     const float* __restrict__ Wa, const float* __restrict__ Wb, int K, int kb, int lane);

// tile_bwd_input's vector branch (N % 4 == 0); NC: N as a compile-time constant (0: take Nr), as tile_fwd_vec's KC
template <int NC>
__device__ __forceinline__ f32x4 tile_bwd_input_vec(f32x4 acc, const float* zrow, int Nr, const float* __restrict__ W, int K, int kcol, bool k_ok, int q) {
  const int N = NC ? NC : Nr;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  {
    for (int np = 0; np < N; np += 16 * kMaxSteps) {
      float wb[kMaxSteps][4];
#pragma unroll
      for (int c = 0; c < kMaxSteps; ++c) {
        if (np + 16 * c >= N) break;                          // (wave-uniform)
        const int n = np + 16 * c + 4 * q;
        const int ns = n < N ? n : 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) wb[c][e] = W[(size_t)(ns + e) * K + kcol];
      }
      auto zread = [&](int c) {                               // (two steps ahead, as tile_fwd's X operand)
        const int n = np + 16 * c + 4 * q;
        const bool ok = n < N;
        const f32x4 za = *reinterpret_cast<const f32x4*>(zrow + (ok ? n : 0));
        return ok ? za : zero;
      };
      f32x4 zq[2] = {zread(0), np + 16 < N ? zread(1) : zero};
#pragma unroll
      for (int c = 0; c < kMaxSteps; ++c) {
        if (np + 16 * c >= N) break;
        const f32x4 dz = zq[c & 1];
        if (c + 2 < kMaxSteps && np + 16 * (c + 2) < N) zq[c & 1] = zread(c + 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma16(dz[e], k_ok ? wb[c][e] : 0.0f, acc);      // (as lin_bwd_input_kernel: the weight of a padded n is whatever row 0 holds, times an exact zero)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  return acc;
}

// acc += dZ . W[:, kb .. kb + 15] for the slab dZ [16][N] in LDS (already multiplied by the activation's derivative).
// The order of lin_bwd_input_kernel: n0 = 0, 16, ...; e = 0..3; the MFMA adds n = n0 + 4q + e over q.  Calling it again
// with another layer's dZ / W continues the same accumulator (the summed gradient of an input two layers share).
__device__ __forceinline__ f32x4 tile_bwd_input(f32x4 acc, const float* dZs, int ldz, int N, const float* __restrict__ W, int K,
                                                int kb, int lane) {
  const int r = lane & 15, q = lane >> 4;
  const int kc = kb + r;
  const bool k_ok = kc < K;
  const int kcol = k_ok ? kc : 0;
  const float* zrow = dZs + r * ldz;
  if ((N & 3) == 0) {
    acc = tile_bwd_input_vec<0>(acc, zrow, N, W, K, kcol, k_ok, q);
  } else {
    for (int n0 = 0; n0 < N; n0 += 16) {
      const int n = n0 + 4 * q;
      float dz[4], wb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dz[e] = n + e < N ? zrow[n + e] : 0.0f;
        wb[e] = (n + e < N && k_ok) ? W[(size_t)(n + e) * K + kc] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma16(dz[e], wb[e], acc);
    }
  }
  return acc;
}

__device__ __forceinline__ f32x4 tile_bwd_input_pair(f32x4 acc, const float* dZa, const float* dZb, int ldz, int N,
                                                     const float* __restrict__ Wa, const float* __restrict__ Wb, int K, int kb, int lane) {
  if ((N & 3) != 0 || N > 16 * kMaxSteps) return tile_bwd_input(tile_bwd_input(acc, dZa, ldz, N, Wa, K, kb, lane), dZb, ldz, N, Wb, K, kb, lane);
  const int r = lane & 15, q = lane >> 4;
  const int kc = kb + r;
  const bool k_ok = kc < K;
  const int kcol = k_ok ? kc : 0;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  float wa[kMaxSteps][4], wb[kMaxSteps][4];
#pragma unroll
  for (int c = 0; c < kMaxSteps; ++c) {
    if (16 * c >= N) break;                                   // (wave-uniform)
    const int n = 16 * c + 4 * q;
    const int ns = n < N ? n : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { wa[c][e] = Wa[(size_t)(ns + e) * K + kcol]; wb[c][e] = Wb[(size_t)(ns + e) * K + kcol]; }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const float* zrow = (half ? dZb : dZa) + r * ldz;
    auto zread = [&](int c) {
      const int n = 16 * c + 4 * q;
      const bool ok = n < N;
      const f32x4 za = *reinterpret_cast<const f32x4*>(zrow + (ok ? n : 0));
      return ok ? za : zero;
    };
    f32x4 zq[2] = {zread(0), 16 < N ? zread(1) : zero};
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (16 * c >= N) break;
      const f32x4 dz = zq[c & 1];
      if (c + 2 < kMaxSteps && 16 * (c + 2) < N) zq[c & 1] = zread(c + 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma16(dz[e], k_ok ? (half ? wb[c][e] : wa[c][e]) : 0.0f, acc);
    }
  }
  return acc;
}

// ---- weight IMAGES for the square H x H layers (H % 16 == 0) ---------------------------------------------------------------
// A slab kernel streams a layer's whole weight matrix through ONE compute unit, and from nn.Linear's row-major layout a
// wave-wide 16-byte load touches sixteen 64-byte pieces of sixteen rows: 34 GB/s per CU, 8-9 us per 256 x 256 layer
// (tools/probe_sac_stages.py).  An image stores, for every (output tile, reduction step), the MFMA B-operand exactly as the
// 64 lanes hold it — one contiguous 1 KiB block per wave-wide load.  Two images per matrix: the forward operand
// (lane (r, q) holds W[16t + r][16c + 4q .. + 3]) and the input-gradient operand (W[16c + 4q .. + 3][16t + r]).  The
// weight-gradient kernel writes both next to the parameter it has just updated; gymrl_sac_pack_images rebuilds them after
// anything else touched the parameters.  Same values in the same registers: the tiles stay bit-identical.
__host__ __device__ __forceinline__ size_t img_fwd_index(int n, int k, int steps) {
  return ((size_t)((n >> 4) * steps + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3);
}
__host__ __device__ __forceinline__ size_t img_bwd_index(int n, int k, int steps) {
  return ((size_t)((k >> 4) * steps + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (k & 15)) * 4 + (n & 3);
}

// tile_fwd for K == N == 16 * steps from the forward image (K1 == K; same MFMA sequence as tile_fwd's vector branch)
template <int SC>
__device__ __forceinline__ f32x4 tile_fwd_img_t(const float* Xs, int ldx, int steps_r, const float* __restrict__ img, int tile, int lane) {
  const int steps = SC ? SC : steps_r;                       // (SC: the step count as a compile-time constant, as tile_fwd_vec's KC)
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const f32x4* wp = reinterpret_cast<const f32x4*>(img) + (size_t)tile * steps * 64 + lane;
  const float* xrow = Xs + r * ldx + 4 * q;
  for (int cp = 0; cp < steps; cp += kMaxSteps) {
    f32x4 wb[kMaxSteps];
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (cp + c >= steps) break;                             // (wave-uniform)
      wb[c] = wp[(size_t)(cp + c) * 64];
    }
    f32x4 xq[2] = {*reinterpret_cast<const f32x4*>(xrow + 16 * cp), *reinterpret_cast<const f32x4*>(xrow + 16 * (cp + 1 < steps ? cp + 1 : cp))};
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (cp + c >= steps) break;
      const f32x4 x = xq[c & 1];
      if (c + 2 < kMaxSteps && cp + c + 2 < steps) xq[c & 1] = *reinterpret_cast<const f32x4*>(xrow + 16 * (cp + c + 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma16(x[e], wb[c][e], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return acc;
}

__device__ __forceinline__ f32x4 tile_fwd_img(const float* Xs, int ldx, int steps, const float* __restrict__ img, int tile, int lane) {
  return tile_fwd_img_t<0>(Xs, ldx, steps, img, tile, lane);
}

// tile_fwd_x2 from the forward image: two slabs share every weight fragment (the 32-row acting workgroups); each accumulator
// sees exactly tile_fwd_img's sequence.
template <int SC>
__device__ __forceinline__ void tile_fwd_img_x2_t(const float* Xs0, const float* Xs1, int ldx, int steps_r, const float* __restrict__ img,
                                                  int tile, int lane, f32x4& acc0, f32x4& acc1) {
  const int steps = SC ? SC : steps_r;
  const int r = lane & 15, q = lane >> 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  acc0 = zero; acc1 = zero;
  const f32x4* wp = reinterpret_cast<const f32x4*>(img) + (size_t)tile * steps * 64 + lane;
  const float* x0 = Xs0 + r * ldx + 4 * q;
  const float* x1 = Xs1 + r * ldx + 4 * q;
  for (int cp = 0; cp < steps; cp += kMaxSteps) {
    f32x4 wb[kMaxSteps];
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (cp + c >= steps) break;                             // (wave-uniform)
      wb[c] = wp[(size_t)(cp + c) * 64];
    }
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (cp + c >= steps) break;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(x0 + 16 * (cp + c));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(x1 + 16 * (cp + c));
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc0 = mfma16(a0[e], wb[c][e], acc0); acc1 = mfma16(a1[e], wb[c][e], acc1); }
    }
  }
}

// tile_bwd_input for K == N == 16 * steps from the input-gradient image
template <int SC>
__device__ __forceinline__ f32x4 tile_bwd_input_img_t(f32x4 acc, const float* dZs, int ldz, int steps_r, const float* __restrict__ img, int ktile,
                                                      int lane) {
  const int steps = SC ? SC : steps_r;
  const int r = lane & 15, q = lane >> 4;
  const f32x4* wp = reinterpret_cast<const f32x4*>(img) + (size_t)ktile * steps * 64 + lane;
  const float* zrow = dZs + r * ldz + 4 * q;
  for (int cp = 0; cp < steps; cp += kMaxSteps) {
    f32x4 wb[kMaxSteps];
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (cp + c >= steps) break;
      wb[c] = wp[(size_t)(cp + c) * 64];
    }
    f32x4 zq[2] = {*reinterpret_cast<const f32x4*>(zrow + 16 * cp), *reinterpret_cast<const f32x4*>(zrow + 16 * (cp + 1 < steps ? cp + 1 : cp))};
#pragma unroll
    for (int c = 0; c < kMaxSteps; ++c) {
      if (cp + c >= steps) break;
      const f32x4 dz = zq[c & 1];
      if (c + 2 < kMaxSteps && cp + c + 2 < steps) zq[c & 1] = *reinterpret_cast<const f32x4*>(zrow + 16 * (cp + c + 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma16(dz[e], wb[c][e], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return acc;
}

__device__ __forceinline__ f32x4 tile_bwd_input_img(f32x4 acc, const float* dZs, int ldz, int steps, const float* __restrict__ img, int ktile,
                                                    int lane) {
  return tile_bwd_input_img_t<0>(acc, dZs, ldz, steps, img, ktile, lane);
}

// One 16 x 16 tile of dW = dZ^T X over rows [0, B) (dZ [B][ldz] column tile nt, X = [X | X2] column tile kb, all in global
// memory), and the tile's share of the bias gradient: the order of lin_bwd_weight_kernel with one slice — rows
// b = b0 + 4e + q, b0 = 0, 16, ...; e = 0..3; per-lane serial column sums folded (q0 + q1) + (q2 + q3).
// acc[g] = dW[nt*16 + 4q + g][kb + r]; colsum (every lane of a column r) = db[nt*16 + r].
constexpr int kWChunk = 8;      // 16-row steps whose loads are in flight together
__device__ __forceinline__ f32x4 tile_bwd_weight(const float* __restrict__ dZ, int ldz, int N, int nt, const float* __restrict__ X,
                                                 int ldx, const float* __restrict__ X2, int ldx2, int K, int K1, int kb, int B,
                                                 int lane, float& colsum_out) {
  const int r = lane & 15, q = lane >> 4;
  const int n = nt * 16 + r;
  const bool n_ok = n < N;
  const int ns = n_ok ? n : 0;
  const int kc = kb + r;
  const bool k_ok = kc < K;
  const int ks = k_ok ? kc : 0;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float colsum = 0.0f;
  for (int bc0 = 0; bc0 < B; bc0 += 16 * kWChunk) {
    float dy[kWChunk][4], xb[kWChunk][4];
#pragma unroll
    for (int c = 0; c < kWChunk; ++c) {
      if (bc0 + 16 * c >= B) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int b = bc0 + 16 * c + 4 * e + q;
        const size_t bs = (size_t)(b < B ? b : 0);
        dy[c][e] = dZ[bs * ldz + ns];
        xb[c][e] = ks < K1 ? X[bs * ldx + ks] : X2[bs * ldx2 + (ks - K1)];
      }
    }
#pragma unroll
    for (int c = 0; c < kWChunk; ++c) {
      if (bc0 + 16 * c >= B) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = n_ok && bc0 + 16 * c + 4 * e + q < B;
        const float dz = ok ? dy[c][e] : 0.0f;
        colsum += dz;
        acc = mfma16(dz, k_ok ? xb[c][e] : 0.0f, acc);
      }
    }
  }
  colsum += __shfl_xor(colsum, 16, 64);
  colsum += __shfl_xor(colsum, 32, 64);
  colsum_out = colsum;
  return acc;
}

// gymrl_lin_bwd_weight's cut of the batch reduction (lin.hip): one chain up to 512 rows; above that `slices` slices of
// rows_per_slice rows (a multiple of 32) whose partial tiles lin_slice_reduce_kernel adds in eight groups — group g the slices
// g * each .. (g + 1) * each - 1 one after another from +0, then the groups in turn.  B >= 16384 on 64-aligned shapes takes the
// 64 x 64-block kernels (another order: not restated here, the fused step stops at 8192 rows).  The fused step's weight-gradient
// tiles (offpolicy_step.hip sac_dw_body) follow the same cut: one wave per slice, the last one to arrive adds them in this order.
__host__ __device__ __forceinline__ bool bwd_weight_big_shape(int B, int N, int K) { return B >= 16384 && N % 64 == 0 && K % 64 == 0; }
__host__ __device__ __forceinline__ int bwd_weight_slices(int B, int N, int K) {
  if (B <= 512) return 1;
  const int by_rows = (B + 255) / 256;
  if (bwd_weight_big_shape(B, N, K)) {      // 64 x 64 blocks: ~2048 waves, never fewer than 256 rows per slice
    int s = (2048 + (N / 64) * (K / 64) - 1) / ((N / 64) * (K / 64));
    if (s > 512) s = 512;
    return s < by_rows ? s : by_rows;
  }
  const int tiles = ((N + 15) / 16) * ((K + 15) / 16);
  int s = (2048 + tiles - 1) / tiles;
  if (s < 16) s = 16;
  if (s > 256) s = 256;
  return s < by_rows ? s : by_rows;
}
__host__ __device__ __forceinline__ int bwd_weight_rows_per_slice(int B, int slices) { return (((B + slices - 1) / slices) + 31) / 32 * 32; }

// torch.optim.Adam's step on ONE element (optim.hip adam_one with grad_scale = scale = 1 and no clamp: the off-policy
// optimisers that clip by norm keep their own launch).  step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t).
struct AdamScalars { float step_size, bc2_sqrt, omb1, beta2, omb2, eps; };
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& a) {
  float gg = g * 1.0f;
  gg = gg * 1.0f;
  m = m + (gg - m) * a.omb1;
  v = v * a.beta2 + a.omb2 * gg * gg;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
}

}  // namespace lin
}  // namespace gymrl
