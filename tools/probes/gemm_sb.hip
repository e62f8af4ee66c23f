// gemm_sb.hip — PROBE (not in libgymrl_hip.so; `make -C tools/probes`): "split-bf16" variants of the PPO update's 256-wide GEMMs.
//
// gfx950 runs `v_mfma_f32_32x32x16_bf16` at 16 x the rate of the exact-f32 `v_mfma_f32_32x32x2_f32` that csrc/gemm.hip
// is built on (2.5 PFLOP/s against 157 TFLOP/s dense).  A float splits EXACTLY into three bf16-representable pieces by
// truncation — hi = its top 8 significant bits, mid = the next 8, lo = the last 8 (x - hi and x - hi - mid are exact in
// f32) — and a product of two bf16 values is exact in the matrix core's f32 accumulator, so
//
//     a * b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + O(2^-24 |a b|)
//
// SIX bf16 MFMAs reproduce an f32 product to ~1.2e-7 relative (the three dropped terms), accumulated in f32 like any f32
// GEMM, in 6/16 of the exact path's matrix time.  This is an f32-ACCURATE GEMM — measured against float64 beside the exact
// kernels on benign and adversarial inputs (tests/test_gemm_sb_gpu.py) — not a restatement of an fmaf chain: the exact
// kernels of gemm.hip stay the DEFAULT, the ones the CPU oracle pins bit for bit and the ones bench.py's headline line runs;
// this mode is reported as a second, clearly labelled bench line (dtype "f32 (3 x bf16 split products, f32 accumulate)").
//
// At 6/16 of the matrix time these products are as much HBM- as MFMA-bound (forward 256 -> 256 at 262,144 rows: 537 MB of
// operand + result against 90 us of matrix time), so everything is built around coalesced traffic:
//
//   forward / input gradient (gemm_sb_ws_kernel) — weight-stationary: a workgroup keeps a [RED x BN] slice of the weights in
//   LDS for its whole life, ALREADY SPLIT into three bf16 planes in the MFMA operand image ([plane][k / 8][n][8 k]: one
//   conflict-free ds_read_b128 per plane, 32-column tile and 16-deep step; 96 KiB at 256 x 64 or 512 x 32).  Each wave
//   streams 32-row tiles past it in stages of 32 reduction indices: the stage's 4 KiB of activations arrive by fully
//   coalesced 16-byte loads (8 lanes per 128-byte row piece), are split on the VALU — which issues in the shadow of the
//   32-cycle MFMAs of the previous stage — and parked as three bf16 planes in the wave's own double-buffered LDS stage
//   (rows padded to 80 bytes: the operand-layout reads are conflict free).  No barrier inside the row loop.
//   Operands are swapped (instruction A = weights, B = activations): a lane's accumulator quads are four adjacent output
//   columns of one row, 16-byte stores with the bias / tanh / tanh' epilogue on them.
//
//   weight gradient (gemm_sb_tn_kernel) — the reduction runs over the rows, and the bf16 instruction wants 8 CONSECUTIVE
//   reduction indices per lane: both operands are transposed on their way into LDS.  A workgroup owns a 256 x 256 output
//   tile for a slice of the rows (4 waves x 128 x 128 quarter = 256 accumulator registers per lane, as gemm_tn_kernel);
//   per 16-row step it loads dY [16][256] and X [16][256] coalesced, splits them, writes the planes [n][16 m] with 2-byte
//   stores (each value is used for 256 outputs: 3072 matrix cycles per step hide them) into a double-buffered stage.
//   Partial tiles and the fixed-order f64 slice reduction are gemm.hip's (same workspace layout, same tn_reduce order).
#include "../../gymrl_amd/csrc/train_device.hpp"
#include "../../include/gymrl.h"
#include "gemm_sb.h"

namespace {

using namespace gymrl;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int kCUs = 256;
enum { EPI_NONE = 0, EPI_TANH = 1, EPI_TANHBWD = 2 };

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v, v), r, off, 0, 0);
}

// x = hi + mid + lo exactly; each piece has <= 8 significant bits (hi, mid: low half of the f32 encoding zero; lo: the
// remainder, which pack2's truncation to its top half leaves exact whenever x has no bits below 2^-24 |x|)
__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, h);
  m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, m);
  l = __builtin_bit_cast(uint32_t, r2);
}
// two f32 encodings -> one register of two bf16 (their top halves; element 0 in the low half)
__device__ __forceinline__ uint32_t pack2(uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

// 4 consecutive floats -> two registers (4 bf16) per plane
__device__ __forceinline__ void split4(const f32x4& v, u32x2& H, u32x2& M, u32x2& L) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split3(v[e], h[e], m[e], l[e]);
  H = u32x2{pack2(h[0], h[1]), pack2(h[2], h[3])};
  M = u32x2{pack2(m[0], m[1]), pack2(m[2], m[3])};
  L = u32x2{pack2(l[0], l[1]), pack2(l[2], l[3])};
}
// 8 consecutive floats -> one operand register quad (8 bf16) per plane
__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, u32x4& H, u32x4& M, u32x4& L) {
  u32x2 h0, m0, l0, h1, m1, l1;
  split4(lo4, h0, m0, l0);
  split4(hi4, h1, m1, l1);
  H = u32x4{h0[0], h0[1], h1[0], h1[1]};
  M = u32x4{m0[0], m0[1], m1[0], m1[1]};
  L = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

// the six products of one 16-deep step, smallest first
__device__ __forceinline__ f32x16 mac6(const u32x4& wh, const u32x4& wm, const u32x4& wl, const u32x4& xh, const u32x4& xm,
                                       const u32x4& xl, f32x16 acc) {
  acc = mfma_bf16(wm, xm, acc);
  acc = mfma_bf16(wl, xh, acc);
  acc = mfma_bf16(wh, xl, acc);
  acc = mfma_bf16(wm, xh, acc);
  acc = mfma_bf16(wh, xm, acc);
  acc = mfma_bf16(wh, xh, acc);
  return acc;
}

struct SbArgs {
  const float* A; int64_t M;
  const float* W; int ldw;
  float* out;
  const float* bias;
  const float* H;
  int slices;
};

// ------------------------------------------------------------------ forward / input gradient ------
// STATUS: correct and f32-accurate, but only 1.1-1.35 x faster than the exact kernels, so not used by the trainers.  Three
// structures were measured at 262,144 x 256 x 256 (profiles/r03_gemm_sb_ablation.txt): this one — one wave per SIMD, software-
// pipelined — 254 us; eight identical waves (two per SIMD) 283 us; four consumer + four producer waves with one barrier per
// stage 269 us; the exact f32 kernel 285 us.  In all three the parts ADD instead of overlapping: MFMAs + operand reads alone
// 120-155 us (the bf16 matrix pipe at full rate pulls the clock to ~1.6 GHz: 196,608 matrix cycles per SIMD), the operands'
// split another 75-95 us EVEN WHEN IT RUNS ON THE SIMD'S OTHER WAVE — as section 4a found for the f32 MFMA, VALU work is
// paid in matrix time on this hardware — and each of the N / 64 column slices re-splits the same activation rows (three
// bf16 planes of a 256-long weight slice fill LDS at 64 columns), which pre-splitting in the producer kernel would only
// trade for 1.5 x the activation bytes on kernels that are HBM-bound at this matrix rate.
// TRANS_W == false: W is [n][red] (nn.Linear weight, forward);  true: W is [red][n] (input gradient).
// ABL (probe build only, tools/abl_gemm_sb.py; wrong results by design): 1 no MFMAs, 2 no split / LDS parking, 4 no activation
// loads inside the loop, 8 no stores.
// PL (producer-side planes, round 4's bounded experiment): the activations arrive ALREADY split — three bf16 planes
// [plane][M][RED] written by gymrl_split_planes (in a pipeline: by the producing layer's epilogue) — so a stage is six
// 16-byte loads per lane that go to the LDS stage as they are: no split, no VALU between the MFMAs; 6 bytes per element in.
template <int RED, int NT, bool TRANS_W, int EPI, int LDO, int ABL = 0, bool PL = false>
__global__ __launch_bounds__(256) void gemm_sb_ws_kernel(SbArgs p) {
  constexpr int BN = 32 * NT, KG = RED / 8, KS = 32, NSTG = RED / KS, kWaves = 4, kThreads = 256, ROWS = 32;
  constexpr int XCH = 5;                                 // 16-byte chunks per staged row (4 used): 80-byte pitch
  constexpr int XBUF = 3 * ROWS * XCH;                   // u32x4 per stage buffer
  static_assert((3 * KG * BN + kWaves * 2 * XBUF) * 16 + BN * 4 <= 160 * 1024, "weight planes + stages must fit LDS");
  __shared__ u32x4 wpl[3 * KG * BN];                     // [plane][kg][n]: 8 bf16 of W(red = 8 kg .. 8 kg + 7, n)
  __shared__ u32x4 xst[kWaves][2][XBUF];                 // per wave: [buffer][plane][row][chunk]
  __shared__ float sbias[BN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, g = lane >> 5;
  const int S = p.slices, RG = gridDim.x / S, b = blockIdx.x;
  int slice, rg;
  if ((RG & 7) == 0) { const int xcd = b & 7, r = b >> 3; slice = r % S; rg = (r / S) * 8 + xcd; }
  else { slice = b % S; rg = b / S; }
  const int n0 = slice * BN;

  const int64_t M = p.M;
  const int64_t tasks = (M + ROWS - 1) / ROWS;
  const int64_t bt_count = (tasks + kWaves - 1) / kWaves;
  auto rows_of = [&](int64_t task) {
    int64_t r = M - task * ROWS;
    return (uint32_t)(r < 0 ? 0 : (r > ROWS ? ROWS : r));
  };
  // stage loads: float4 number q * 64 + lane of the [32 rows][32 k] stage tile = row (q * 64 + lane) / 8, floats 4 * (lane % 8)
  const int srow = lane >> 3, sc4 = lane & 7;
  const uint32_t soff = (uint32_t)(srow * RED + 4 * sc4) * 4u;          // + q * 8 rows, + stage * 32 floats
  auto fetch = [&](__amdgpu_buffer_rsrc_t r, int stg, f32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ld4(r, soff + (uint32_t)(q * 8 * RED + stg * KS) * 4u);
  };
  u32x4* myst = &xst[wave][0][0];
  // park: the lane's four float4 (rows srow + 8 q, k-quad sc4) -> halves of the 16-byte operand chunks of three planes
  auto park = [&](int buf, const f32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x2 H, Mm, L;
      split4(v[q], H, Mm, L);
      const int row = srow + 8 * q;
      u32x2* base = reinterpret_cast<u32x2*>(myst + buf * XBUF + row * XCH + (sc4 >> 1)) + (sc4 & 1);
      base[0 * ROWS * XCH * 2] = H;
      base[1 * ROWS * XCH * 2] = Mm;
      base[2 * ROWS * XCH * 2] = L;
    }
  };

  // The activations of a whole row tile are in flight at any time: ring slot s holds stage s of a tile as raw f32 (4 x 16
  // bytes per lane); the moment a slot has been split and parked in LDS it is refilled with the same stage of the NEXT tile, so
  // every load has a full tile of matrix work (NSTG stages, ~3 us) to arrive — one stage ahead (0.35 us) exposed the whole
  // HBM latency on every stage (measured: 5 x the matrix time).
  int64_t bt = rg;
  constexpr int NLD = PL ? 6 : 4;                        // 16-byte loads per lane and stage
  f32x4 ring[NSTG][NLD];
  auto tile_rsrc = [&](int64_t bt_) {
    const int64_t task = bt_ * kWaves + wave;
    return rsrc_of(p.A + task * ROWS * RED, rows_of(task) * RED * 4u);
  };
  // PL: plane pl of a row tile; lane (l >> 2, l & 3) loads row 16 hq + (l >> 2), reduction indices 8 (l & 3) .. + 7 of the stage
  const uint16_t* planes = reinterpret_cast<const uint16_t*>(p.A);
  auto plane_rsrc = [&](int64_t bt_, int pl) {
    const int64_t task = bt_ * kWaves + wave;
    return rsrc_of(planes + ((int64_t)pl * M + task * ROWS) * RED, rows_of(task) * RED * 2u);
  };
  const uint32_t poff = (uint32_t)((lane >> 2) * RED + 8 * (lane & 3)) * 2u;
  auto fetch_pl = [&](int64_t bt_, int stg, f32x4 (&v)[NLD]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const __amdgpu_buffer_rsrc_t r = plane_rsrc(bt_, pl);
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) v[2 * pl + hq] = ld4(r, poff + (uint32_t)(hq * 16 * RED + stg * KS) * 2u);
    }
  };
  auto park_pl = [&](int buf, const f32x4 (&v)[NLD]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int hq = 0; hq < 2; ++hq)
        myst[buf * XBUF + pl * ROWS * XCH + (16 * hq + (lane >> 2)) * XCH + (lane & 3)] = __builtin_bit_cast(u32x4, v[2 * pl + hq]);
  };
  if constexpr (PL) {
#pragma unroll
    for (int stg = 0; stg < NSTG; ++stg) fetch_pl(bt, stg, ring[stg]);
  } else {
    const __amdgpu_buffer_rsrc_t first = tile_rsrc(bt);
#pragma unroll
    for (int stg = 0; stg < NSTG; ++stg) {
      f32x4 (&dst)[4] = reinterpret_cast<f32x4 (&)[4]>(ring[stg]);
      fetch(first, stg, dst);                             // in flight under the weight fill
    }
  }

  // ---- weight slice -> three bf16 planes in LDS, once per workgroup
  for (int idx = tid; idx < BN * KG; idx += kThreads) {
    const int n = idx % BN, kg = idx / BN;
    f32x4 w0, w1;
    if constexpr (!TRANS_W) {
      const float* src = p.W + (size_t)(n0 + n) * p.ldw + 8 * kg;
      w0 = *reinterpret_cast<const f32x4*>(src);
      w1 = *reinterpret_cast<const f32x4*>(src + 4);
    } else {
      const float* src = p.W + (size_t)(8 * kg) * p.ldw + n0 + n;
      const size_t s = (size_t)p.ldw;
      w0 = f32x4{src[0], src[s], src[2 * s], src[3 * s]};
      w1 = f32x4{src[4 * s], src[5 * s], src[6 * s], src[7 * s]};
    }
    u32x4 H, Mm, L;
    split8(w0, w1, H, Mm, L);
    wpl[(0 * KG + kg) * BN + n] = H;
    wpl[(1 * KG + kg) * BN + n] = Mm;
    wpl[(2 * KG + kg) * BN + n] = L;
  }
  if (tid < BN) sbias[tid] = (EPI != EPI_TANHBWD && p.bias) ? p.bias[n0 + tid] : 0.0f;
  __syncthreads();

  const uint32_t ooff = (uint32_t)(j * LDO + 4 * g) * 4u;               // lane (j, g): row j, columns 8 q + 4 g .. + 3 of a tile
  static_assert(NSTG % 2 == 0, "stage buffers alternate across tiles");
  if constexpr (PL) {
    park_pl(0, ring[0]);
    fetch_pl(bt + RG, 0, ring[0]);
  } else {
    f32x4 (&r0)[4] = reinterpret_cast<f32x4 (&)[4]>(ring[0]);
    park(0, r0);
    fetch(tile_rsrc(bt + RG), 0, r0);
  }
  // operand registers of one 16-deep step: three planes of the activations and of NT weight tiles.  Two sets: the set of a
  // step is requested from LDS while the previous step's 6 NT MFMAs run (left to the compiler, every read sat next to its
  // first use behind an lgkmcnt(0): the LDS latency was exposed ~8 times per stage)
  struct Ops { u32x4 xh, xm, xl, wh[NT], wm[NT], wl[NT]; };
  auto load_ops = [&](Ops& o, int buf, int stg, int ss) {
    const u32x4* xs = myst + buf * XBUF + j * XCH + 2 * ss + g;
    o.xh = xs[0 * ROWS * XCH]; o.xm = xs[1 * ROWS * XCH]; o.xl = xs[2 * ROWS * XCH];
    const int kg = stg * 4 + 2 * ss + g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      o.wh[nt] = wpl[(0 * KG + kg) * BN + 32 * nt + j];
      o.wm[nt] = wpl[(1 * KG + kg) * BN + 32 * nt + j];
      o.wl[nt] = wpl[(2 * KG + kg) * BN + 32 * nt + j];
    }
  };
  Ops o0, o1;
  load_ops(o0, 0, 0, 0);
  f32x4 bv[NT][4];                                      // this lane's bias columns (epilogue), read once
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[nt][q] = *reinterpret_cast<const f32x4*>(&sbias[32 * nt + 8 * q + 4 * g]);
#pragma unroll 1
  for (; bt < bt_count; bt += RG) {
    const int64_t task = bt * kWaves + wave;
    const uint32_t rows = rows_of(task);
    const __amdgpu_buffer_rsrc_t n1 = tile_rsrc(bt + RG), n2 = tile_rsrc(bt + 2 * RG);   // past the end: zero rows, loads return 0
    const __amdgpu_buffer_rsrc_t co = rsrc_of(p.out + task * ROWS * LDO + n0, rows * LDO * 4u - (rows ? n0 * 4u : 0u));
    f32x4 hv[NT][4];
    if constexpr (EPI == EPI_TANHBWD) {
      const __amdgpu_buffer_rsrc_t chh = rsrc_of(p.H + task * ROWS * LDO + n0, p.H ? rows * LDO * 4u - (rows ? n0 * 4u : 0u) : 0u);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[nt][q] = ld4(chh, ooff + (uint32_t)((32 * nt + 8 * q) * 4));
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
    for (int stg = 0; stg < NSTG; ++stg) {
      const int buf = stg & 1;
      load_ops(o1, buf, stg, 1);                        // second half of this stage, requested under the first half's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      // The stage's 12 NT MFMAs, each followed — pinned — by a slice of the work that prepares the NEXT stage (this tile's, or
      // stage 0 of the next tile; its ring slot then starts over one tile ahead).  A wave issues in order: an MFMA issued
      // right behind another waits ~28 cycles for the matrix pipe with the issue port blocked, while ~7 VALU instructions
      // fit into that wait for free — so the split's ~100 VALU instructions, the LDS writes and the refills are spread over
      // the first three quarters of the MFMAs (measured with them bunched: pipe 41 % busy, a third of the wave's cycles in
      // issue stalls), and the next stage's first operands are requested behind the last slice, a quarter stage ahead of
      // their use.  12 slices: per float4 of the lane (q) two slices of two element splits and one of packing + the three
      // LDS writes + the slot's refill.
      const int ns = (stg + 1) % NSTG;
      {
        constexpr int TOT = 12 * NT, PK = TOT * 3 / 4;
        uint32_t ph[4], pm[4], pl[4];
        u32x4* const dstbuf = myst + (buf ^ 1) * XBUF;
        auto slice = [&](int c) {
          if constexpr (PL) {                             // six pieces: the loaded chunk goes to the LDS stage as it is, its slot is refilled
            if (c < 6) {
              const int pl = c >> 1, hq = c & 1;
              dstbuf[pl * ROWS * XCH + (16 * hq + (lane >> 2)) * XCH + (lane & 3)] = __builtin_bit_cast(u32x4, ring[ns][c]);
              const int64_t nbt = ns ? bt + RG : bt + 2 * RG;
              ring[ns][c] = ld4(plane_rsrc(nbt, pl), poff + (uint32_t)(hq * 16 * RED + ns * KS) * 2u);
            }
            return;
          }
          const int q = c / 3, part = c % 3;
          if (part < 2) {
            if constexpr (!(ABL & 2)) {
#pragma unroll
              for (int e = 2 * part; e < 2 * part + 2; ++e) split3(ring[ns][q][e], ph[e], pm[e], pl[e]);
            }
          } else {
            if constexpr (!(ABL & 2)) {
              const int row = srow + 8 * q;
              u32x2* base = reinterpret_cast<u32x2*>(dstbuf + row * XCH + (sc4 >> 1)) + (sc4 & 1);
              base[0 * ROWS * XCH * 2] = u32x2{pack2(ph[0], ph[1]), pack2(ph[2], ph[3])};
              base[1 * ROWS * XCH * 2] = u32x2{pack2(pm[0], pm[1]), pack2(pm[2], pm[3])};
              base[2 * ROWS * XCH * 2] = u32x2{pack2(pl[0], pl[1]), pack2(pl[2], pl[3])};
            } else {
              asm volatile("" :: "v"(ring[ns][q]));
            }
            if constexpr (!(ABL & 4)) ring[ns][q] = ld4(ns ? n1 : n2, soff + (uint32_t)(q * 8 * RED + ns * KS) * 4u);
          }
        };
#pragma unroll
        for (int i = 0; i < TOT; ++i) {
          const int half = i / (6 * NT), k = i % (6 * NT);
          const int pr = k / NT, nt = k % NT;           // product pr of mac6's order on accumulator nt
          const Ops& o = half ? o1 : o0;
          const u32x4& w = (pr == 1) ? o.wl[nt] : ((pr == 0 || pr == 3) ? o.wm[nt] : o.wh[nt]);
          const u32x4& x = (pr == 2) ? o.xl : ((pr == 0 || pr == 4) ? o.xm : o.xh);
          if constexpr (!(ABL & 1)) acc[nt] = mfma_bf16(w, x, acc[nt]);
          else acc[nt][0] += __builtin_bit_cast(float, w[0] ^ x[1]);
          if (i < PK) {
#pragma unroll
            for (int c = i * 12 / PK; c < (i + 1) * 12 / PK; ++c) slice(c);
          }
          if (i == PK - 1) {
            __builtin_amdgcn_sched_barrier(0);
            load_ops(o0, buf ^ 1, ns, 0);               // first half of the next stage (o0 was last read by MFMA 6 NT - 1)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // epilogue: lane (j, g) holds row j, columns n0 + 32 nt + 8 q + 4 g .. + 3 in accumulator quad q
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = acc[nt][4 * q + e];
          if constexpr (EPI == EPI_NONE) v[e] = z + bv[nt][q][e];
          if constexpr (EPI == EPI_TANH) v[e] = train_tanhf(z + bv[nt][q][e]);
          if constexpr (EPI == EPI_TANHBWD) v[e] = p.H ? z * (1.0f - hv[nt][q][e] * hv[nt][q][e]) : z;
        }
        if constexpr (!(ABL & 8)) st4(co, ooff + (uint32_t)((32 * nt + 8 * q) * 4), v);
        else asm volatile("" :: "v"(v));
      }
  }
}

#ifdef GYMRL_PROF_BUILD
int g_sb_abl = 0;
#endif
// ---- the three bf16 planes of an f32 array (what a producing layer's epilogue would write): P[plane][n] -------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const f32x4* __restrict__ x, int64_t n4, u32x2* __restrict__ planes) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    u32x2 H, Mm, L;
    split4(x[i], H, Mm, L);
    planes[i] = H; planes[n4 + i] = Mm; planes[2 * n4 + i] = L;
  }
}

template <int EPI, int LDO>
void launch_sb_ws_planes(const SbArgs& a, hipStream_t s) {
  const int64_t tasks = (a.M + 31) / 32, bts = (tasks + 3) / 4;
  int64_t rg = kCUs / a.slices;
  if (bts < rg) rg = bts;
  if (rg < 1) rg = 1;
  hipLaunchKernelGGL((gemm_sb_ws_kernel<256, 2, false, EPI, LDO, 0, true>), dim3((unsigned)(rg * a.slices)), dim3(256), 0, s, a);
}

template <int RED, int NT, bool TRANS_W, int EPI, int LDO>
void launch_sb_ws(const SbArgs& a, hipStream_t s) {
  const int64_t tasks = (a.M + 31) / 32, bts = (tasks + 3) / 4;
  int64_t rg = kCUs / a.slices;
  if (bts < rg) rg = bts;
  if (rg < 1) rg = 1;
  const dim3 grid((unsigned)(rg * a.slices)), block(256);
#ifdef GYMRL_PROF_BUILD
  if constexpr (EPI == EPI_NONE && LDO == 256) {
    switch (g_sb_abl) {
      case 1: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 1>), grid, block, 0, s, a); return;
      case 2: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 2>), grid, block, 0, s, a); return;
      case 3: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 3>), grid, block, 0, s, a); return;
      case 4: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 4>), grid, block, 0, s, a); return;
      case 6: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 6>), grid, block, 0, s, a); return;
      case 8: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 8>), grid, block, 0, s, a); return;
      case 12: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 12>), grid, block, 0, s, a); return;
      case 14: hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO, 14>), grid, block, 0, s, a); return;
      default: break;
    }
  }
#endif
  hipLaunchKernelGGL((gemm_sb_ws_kernel<RED, NT, TRANS_W, EPI, LDO>), grid, block, 0, s, a);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

#ifdef GYMRL_PROF_BUILD
int gymrl_gemm_sb_config(int value) { g_sb_abl = value; return 0; }   // probe build only (tools/abl_gemm_sb.py)
#endif

int gymrl_linear_fwd_sb(const float* X, const float* W, const float* b, int64_t B, int K, int N, int act, float* Y,
                        void* stream) {
  if (!X || !W || !Y || B < 0 || K != 256 || (N != 256 && N != 512) || (act != 0 && act != 1) || !al16(X) || !al16(W) || !al16(Y))
    return -22;
  if (B == 0) return 0;
  SbArgs a{};
  a.A = X; a.M = B; a.W = W; a.ldw = K; a.out = Y; a.bias = b; a.slices = N / 64;
  hipStream_t s = (hipStream_t)stream;
  if (N == 256) {
    if (act) launch_sb_ws<256, 2, false, EPI_TANH, 256>(a, s); else launch_sb_ws<256, 2, false, EPI_NONE, 256>(a, s);
  } else {
    if (act) launch_sb_ws<256, 2, false, EPI_TANH, 512>(a, s); else launch_sb_ws<256, 2, false, EPI_NONE, 512>(a, s);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_split_planes(const float* X, int64_t n, void* planes, void* stream) {
  if (!X || !planes || n < 0 || n % 4 || !al16(X) || !al16(planes)) return -22;
  if (n == 0) return 0;
  const int64_t n4 = n / 4, want = (n4 + 255) / 256;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(X), n4, reinterpret_cast<u32x2*>(planes));
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_fwd_sb_planes(const void* X_planes, const float* W, const float* b, int64_t B, int K, int N, int act, float* Y,
                               void* stream) {
  if (!X_planes || !W || !Y || B < 0 || K != 256 || (N != 256 && N != 512) || (act != 0 && act != 1) || !al16(X_planes) || !al16(W) ||
      !al16(Y))
    return -22;
  if (B == 0) return 0;
  SbArgs a{};
  a.A = static_cast<const float*>(X_planes); a.M = B; a.W = W; a.ldw = K; a.out = Y; a.bias = b; a.slices = N / 64;
  hipStream_t s = (hipStream_t)stream;
  if (N == 256) {
    if (act) launch_sb_ws_planes<EPI_TANH, 256>(a, s); else launch_sb_ws_planes<EPI_NONE, 256>(a, s);
  } else {
    if (act) launch_sb_ws_planes<EPI_TANH, 512>(a, s); else launch_sb_ws_planes<EPI_NONE, 512>(a, s);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_bwd_input_sb(const float* dY, const float* W, const float* H, int64_t B, int N, int K, float* dX,
                              void* stream) {
  if (!dY || !W || !dX || B < 0 || K != 256 || (N != 256 && N != 512) || !al16(dY) || !al16(W) || !al16(dX) || (H && !al16(H)))
    return -22;
  if (B == 0) return 0;
  SbArgs a{};
  a.A = dY; a.M = B; a.W = W; a.ldw = K; a.out = dX; a.H = H;
  hipStream_t s = (hipStream_t)stream;
  if (N == 256) {
    a.slices = 4;
    launch_sb_ws<256, 2, true, EPI_TANHBWD, 256>(a, s);
  } else {
    a.slices = 8;
    launch_sb_ws<512, 1, true, EPI_TANHBWD, 256>(a, s);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
