// gymrl_device.hpp — device-side helpers shared by the gfx950 kernels.
//
//  * bit-reproducible f32 exp/log/sincos/tanh built only from IEEE +,*,fma,
//    rint, frexp, ldexp (no ocml transcendentals), so a kernel result can be
//    bit-compared with the CPU oracle's independent restatement;
//  * Philox4x32-10 counter-based RNG (integer only => exact on every machine);
//  * wave64 / workgroup reductions.
//
// All translation units are compiled with -ffp-contract=off: every fused
// multiply-add in this tree is an explicit fmaf()/fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GYMRL_WAVE 64

#define GYMRL_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return -1000 - (int)e__;            \
  } while (0)

namespace gymrl {

// ---------------------------------------------------------------- Philox ---
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32(uint64_t key, uint32_t c0, uint32_t c1,
                                            uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

// [0,1) with 24 random bits
__device__ __forceinline__ float u01f(uint32_t x) { return (float)(x >> 8) * 0x1p-24f; }
// (0,1] with 24 random bits (safe for log)
__device__ __forceinline__ float u01f_open0(uint32_t x) { return (float)((x >> 8) + 1u) * 0x1p-24f; }
// [0,1) with 53 random bits
__device__ __forceinline__ double u01d(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}

// RNG domain tags (c3 of the counter)
enum : uint32_t {
  RNG_ENV_RESET = 0x10000000u,
  RNG_ENV_STEP  = 0x20000000u,
  RNG_POLICY    = 0x30000000u,
  RNG_REPLAY    = 0x40000000u,
  RNG_NOISE     = 0x50000000u,
  RNG_SHUFFLE   = 0x60000000u,
};

// ------------------------------------------------------ keyed permutation ---
// A bijection of [0, M) evaluated per element: 6 alternating Feistel rounds on ceil(log2 M) bits (low half a
// bits, high half b bits, round function = Philox4x32-10 keyed by (seed, counter)), cycle-walked back into
// [0, M) (at most 2 evaluations on average since 2^bits < 2M).  No sort, no table, nothing read from memory.
__device__ __forceinline__ uint32_t feistel_once(uint32_t x, int a, int b, uint64_t seed, uint64_t counter) {
  const uint32_t mask_lo = (1u << a) - 1u, mask_hi = (1u << b) - 1u;
  uint32_t lo = x & mask_lo, hi = x >> a;
  const uint32_t c2 = (uint32_t)counter, c3 = RNG_SHUFFLE | ((uint32_t)(counter >> 32) & 0x0FFFFFFFu);
#pragma unroll
  for (uint32_t r = 0; r < 6; ++r) {
    if ((r & 1u) == 0u) lo ^= philox4x32(seed, hi, r, c2, c3).x & mask_lo;
    else                hi ^= philox4x32(seed, lo, r, c2, c3).x & mask_hi;
  }
  return (hi << a) | lo;
}
__device__ __forceinline__ uint32_t keyed_permute(uint32_t i, uint32_t M, int a, int b, uint64_t seed, uint64_t counter) {
  uint32_t x = feistel_once(i, a, b, seed, counter);
  while (x >= M) x = feistel_once(x, a, b, seed, counter);
  return x;
}

// ------------------------------------------------- reproducible f32 math ---
// exp: Cody-Waite reduction by ln2 (hi/lo), degree-6 polynomial (Cephes expf
// coefficients), result flushed to 0 below -87.33 and to +inf above 88.72.
__device__ __forceinline__ float det_expf(float x) {
  if (!(x > -87.33654f)) return (x != x) ? x : 0.0f;
  if (x > 88.72283f) return __builtin_inff();
  float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, r * r, r) + 1.0f;
  return __builtin_ldexpf(y, (int)n);
}

// log for finite x > 0 (normal or denormal); Cephes logf polynomial.
__device__ __forceinline__ float det_logf(float x) {
  int e;
  float m = __builtin_frexpf(x, &e);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = fmaf(p, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = p * m * z;
  float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(-0.5f, z, y);
  float r = m + y;
  return fmaf(fe, 0.693359375f, r);
}

// sin & cos for |x| < ~8000: quadrant reduction with a 3-term Cody-Waite pi/2,
// Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void det_sincosf(float x, float* s, float* c) {
  float q = __builtin_rintf(x * 0.636619772367581343f);  // 2/pi
  int qi = (int)q;
  float r = fmaf(q, -1.5703125f, x);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188e-8f, r);
  float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float sn = fmaf(ps * z, r, r);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float cs = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  float ss = (qi & 1) ? cs : sn;
  float cc = (qi & 1) ? sn : cs;
  if (qi & 2) ss = -ss;
  if ((qi + 1) & 2) cc = -cc;
  *s = ss; *c = cc;
}

// Reproducible float64 sin/cos for the classic-control steppers (gymnasium keeps CartPole / Pendulum state in float64):
// quadrant by rint(x * 2/pi), two-constant Cody-Waite reduction with fma (pi/2 = 33 leading bits + tail, fdlibm's pio2_1 /
// pio2_1t: q * pio2_1 is exact for |q| < 2^20, i.e. |x| < 1.6e6 — Pendulum's angle stays below pi + 200 * 0.4), fdlibm's
// degree-13 / degree-12 minimax kernels on |r| <= pi/4 as fma Horner chains.  <= 1.6 ulp against sinl / cosl over
// [-100, 100] (tools/check_det_sincos.c).  Every operation is spelled out (fma only where written), so a host
// restatement of the same lines — the test suite's CPU checker has one — agrees bit for bit, which ocml's and libm's
// sin / cos do not.
__device__ __forceinline__ void det_sincos(double x, double* s, double* c) {
  const double q = __builtin_rint(x * 0.63661977236758134308);
  const long long qi = (long long)q;
  double r = __builtin_fma(q, -1.57079632673412561417e+00, x);
  r = __builtin_fma(q, -6.07710050650619224932e-11, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  const double sn = __builtin_fma(ps * z, r, r);
  double pc = -1.13596475577881948265e-11;
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  const double cs = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
  double ss = (qi & 1) ? cs : sn;
  double cc = (qi & 1) ? sn : cs;
  if (qi & 2) ss = -ss;
  if ((qi + 1) & 2) cc = -cc;
  *s = ss; *c = cc;
}

// tanh via det_expf; odd polynomial below 0.625 (Cephes tanhf).
__device__ __forceinline__ float det_tanhf(float x) {
  float a = __builtin_fabsf(x);
  if (a >= 0.625f) {
    float r;
    if (a > 9.0f) r = 1.0f;
    else { float e = det_expf(a + a); r = 1.0f - 2.0f / (e + 1.0f); }
    return x < 0.0f ? -r : r;
  }
  float z = x * x;
  float p = -5.70498872745e-3f;
  p = fmaf(p, z, 2.06390887954e-2f);
  p = fmaf(p, z, -5.37397155531e-2f);
  p = fmaf(p, z, 1.33314422036e-1f);
  p = fmaf(p, z, -3.33332819422e-1f);
  return fmaf(p * z, x, x);
}

// Branch-free det_tanhf (bit-identical for every finite x): both ranges are evaluated and
// selected, so a wave whose lanes straddle |x| = 0.625 does not execute two divergent paths
// one after the other.  Used where tanh sits in an MFMA epilogue.
__device__ __forceinline__ float det_tanhf_sel(float x) {
  const float a = __builtin_fabsf(x);
  const float t = __builtin_fminf(a, 9.0f);          // keeps exp in range; overridden below when a > 9
  const float xx = t + t;
  float n = __builtin_rintf(xx * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, xx);
  r = fmaf(n, 2.12194440e-4f, r);
  float pe = 1.9875691500e-4f;
  pe = fmaf(pe, r, 1.3981999507e-3f);
  pe = fmaf(pe, r, 8.3334519073e-3f);
  pe = fmaf(pe, r, 4.1665795894e-2f);
  pe = fmaf(pe, r, 1.6666665459e-1f);
  pe = fmaf(pe, r, 5.0000001201e-1f);
  const float e = __builtin_ldexpf(fmaf(pe, r * r, r) + 1.0f, (int)n);
  float big = 1.0f - 2.0f / (e + 1.0f);
  big = a > 9.0f ? 1.0f : big;
  big = x < 0.0f ? -big : big;
  const float z = x * x;
  float p = -5.70498872745e-3f;
  p = fmaf(p, z, 2.06390887954e-2f);
  p = fmaf(p, z, -5.37397155531e-2f);
  p = fmaf(p, z, 1.33314422036e-1f);
  p = fmaf(p, z, -3.33332819422e-1f);
  const float small = fmaf(p * z, x, x);
  return a >= 0.625f ? big : small;
}

// --------------------------------------------------- online GAE chunk maps ---
// One step of the forward composition used by gymrl_gae(variant 2); shared by the
// categorical-sample kernel (fused) and the flush kernel.  running = f64[2][N].
__device__ __forceinline__ void gae_online_compose(float r, uint8_t d, float v_prev, float v_cur, double gamma,
                                                   double gl, int first, int last, double* __restrict__ running,
                                                   double2* __restrict__ agg_row, int N, int i) {
  double A = first ? 1.0 : running[i];
  double b = first ? 0.0 : running[(size_t)N + i];
  const double nd = 1.0 - (double)(d != 0);
  const double delta = ((double)r + (gamma * (double)v_cur) * nd) - (double)v_prev;
  b = b + A * delta;
  A = A * (gl * nd);
  if (last) agg_row[i] = make_double2(A, b);
  else { running[i] = A; running[(size_t)N + i] = b; }
}

// The blocked scan's second pass for ONE env, by one lane: carry[c][n] = the advantage entering chunk c from the future.  It
// restates gae.hip's gae_blk_carry_kernel operation for operation — rounds of 16 segments x 8 chunks walked from the last
// chunk down, a segment's maps composed top-down into one (A, b), the value entering a segment folded through the HIGHER
// segments' composed maps, the chunks inside a segment replayed one by one — so the carries (and with them every advantage)
// carry the same bits whether that kernel computes them or the persistent rollout does at its tail (gymrl_gae variant 3).
constexpr int kGaeCarrySeg = 16, kGaeCarryL = 8;     // == gae.hip kCarrySeg / kCarryL
// m: kGaeCarrySeg double2 of scratch for this lane (element s at m[s * m_stride]; LDS in the rollout kernels — as a private
// array it went to the stack of a kernel whose register budget the solver owns).  The segment loops are NOT unrolled for the
// same reason: eight maps in flight per segment are all the memory parallelism a lane needs at a kernel's tail.
__device__ __forceinline__ void gae_carry_scan(const double2* __restrict__ agg, int C, int N, int n, double* __restrict__ carry,
                                               double2* m, int m_stride) {
  double top = 0.0;
  for (int c_hi = C; c_hi > 0; c_hi -= kGaeCarrySeg * kGaeCarryL) {
#pragma unroll 1
    for (int seg = 0; seg < kGaeCarrySeg; ++seg) {
      const int c_top = c_hi - 1 - (kGaeCarrySeg - 1 - seg) * kGaeCarryL;
      double2 ab[kGaeCarryL];
#pragma unroll
      for (int j = 0; j < kGaeCarryL; ++j) {
        const int c = c_top - j;
        ab[j] = c >= 0 ? agg[(size_t)c * N + n] : make_double2(1.0, 0.0);
      }
      double A = 1.0, b = 0.0;
#pragma unroll
      for (int j = 0; j < kGaeCarryL; ++j) { b = ab[j].y + ab[j].x * b; A = ab[j].x * A; }
      m[seg * m_stride] = make_double2(A, b);
    }
    double x_in = top;                                // the value entering segment 15, then 14, ...
#pragma unroll 1
    for (int seg = kGaeCarrySeg - 1; seg >= 0; --seg) {
      const int c_top = c_hi - 1 - (kGaeCarrySeg - 1 - seg) * kGaeCarryL;
      double2 ab[kGaeCarryL];
#pragma unroll
      for (int j = 0; j < kGaeCarryL; ++j) {
        const int c = c_top - j;
        ab[j] = c >= 0 ? agg[(size_t)c * N + n] : make_double2(1.0, 0.0);
      }
      double x = x_in;
#pragma unroll
      for (int j = 0; j < kGaeCarryL; ++j) {
        const int c = c_top - j;
        if (c >= 0) carry[(size_t)c * N + n] = x;
        x = ab[j].y + ab[j].x * x;
      }
      const double2 ms = m[seg * m_stride];
      x_in = ms.y + ms.x * x_in;
    }
    top = x_in;
  }
}

// ------------------------------------------------------------ reductions ---
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-wide sum of K doubles per thread -> K atomics per block (lane 0 of
// wave 0).  BLOCK is the workgroup size (multiple of 64, <= 1024).
template <int K, int BLOCK>
__device__ __forceinline__ void block_atomic_add(double (&v)[K], double* out) {
  __shared__ double sm[K][BLOCK / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = wave_sum(v[k]);
    if (lane == 0) sm[k][wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) s += sm[threadIdx.x][w];
    atomicAdd(out + threadIdx.x, s);
  }
}

// NoisyLinear's factorised noise (rainbow_dqn_cartpole.py:77-87): f(x) = sign(x) sqrt(|x|) of N(0,1) draws, Box-Muller on
// Philox(seed, counter; stream 0 = input side, 1 = output side, element index).  Shared by gymrl_noisy_noise
// (offpolicy.hip) and the fused head's gymrl_noisy_combine (lin.hip), which must produce the same bits.
__device__ __forceinline__ float scale_noise(float x) {            // x.sign().mul(x.abs().sqrt())
  const float s = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
  return s * sqrtf(fabsf(x));
}
__device__ __forceinline__ float box_muller(uint64_t seed, uint64_t counter, uint32_t stream, uint32_t i) {
  const u32x4 r = philox4x32(seed, i, stream, (uint32_t)counter, RNG_NOISE | (uint32_t)((counter >> 32) & 0x0FFFFFFFu));
  const float u1 = u01f_open0(r.x), u2 = u01f(r.y);
  float s, c;
  det_sincosf(6.28318530717958647692f * u2, &s, &c);
  return sqrtf(-2.0f * det_logf(u1)) * c;
}

}  // namespace gymrl
