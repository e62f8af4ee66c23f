// gemm.hip — hand-written f32-MFMA GEMMs of the PPO ActorCritic update (gfx950).
//
// One PPO minibatch (ppo_lunarlander.py:274-307, B = 262,144 rows at BASELINE config 2) is six
// 256-wide contractions: 1.18 MFLOP per row, 310 GFLOP per minibatch — the MFMA-bound 2/3 of the
// update.  They are exact-f32 `v_mfma_f32_32x32x2_f32` kernels (the reference computes in f32; gfx950
// has no TF32) with the elementwise work of the layer fused into the epilogue, so no activation is
// streamed through HBM a second time for tanh, tanh' or a bias-gradient reduction:
//
//   gymrl_linear_fwd         Y = act(X W^T + b)            "weight-stationary", act = tanh in the epilogue
//   gymrl_linear_bwd_input   dX = (dY W) * (1 - H^2), db_below = colsum(dX)   same kernel, W read transposed
//   gymrl_linear_bwd_weight  dW = dY^T X                   rows split over the chip, fixed-order reduction
//
// Weight-stationary (forward / input gradient).  The weight matrix is small (256 x 256 .. 512 x 256 f32)
// and the row count is huge, so a workgroup keeps a [RED x BN] slice of the weights in LDS for its whole
// life (128 KiB: RED x BN = 256 x 128 or 512 x 64) and streams row tiles of the activations past it.  A
// wave owns 32*RT rows x BN columns; its A operand comes straight from HBM/L2 into the MFMA register
// layout (lane (i, h) = lane & 31, lane >> 5 loads the 16 bytes A[row i][8c + 4h .. 8c + 4h + 3] of
// reduction chunk c — nothing is shared between waves, so LDS staging would only add a round trip), its
// B operand is one conflict-free ds_read_b128 per 32-column tile and chunk.  Per chunk of 8 reduction
// indices a wave issues RT*NT*4 MFMAs (64 cycles each) against RT global loads and NT LDS reads: the
// loop is MFMA-issue bound by construction; A is prefetched four chunks (>= 1024 MFMA cycles) ahead and
// across the epilogue into the next row tile.  The column slices of one row group are placed on the same
// XCD (block b runs on XCD b % 8), so the second..fourth read of an activation row is an L2 hit.
//
// Reduction order (documented so that the CPU oracle restates it bit for bit; an f32 MFMA is an fmaf chain):
// within chunk c the products are accumulated in the order 8c+0, 8c+4, 8c+1, 8c+5, 8c+2, 8c+6, 8c+3, 8c+7;
// chunks ascend; the accumulator starts at +0; bias is added after the chain.
//
// Weight gradient.  dW[n][k] = sum_m dY[m][n] X[m][k]: the reduction runs over the rows.  Both operands
// are read from HBM directly in MFMA layout (row pairs m, m+1 are the instruction's K = 2; a lane's
// dwordx4 / dwordx2 covers 4 / 2 adjacent columns, which only permutes which output element a lane owns).
// A workgroup (8 waves) holds a whole 256 x 256 output tile in accumulators (128 registers per lane) for a
// slice of the rows and writes one f32 partial tile; gymrl_linear_bwd_weight's second launch adds the
// partial tiles in a fixed order (f64).  Order: inside a slice rows ascend (fmaf chain from +0); slices
// are summed as ((g0 + g1) + g2) + g3 with g_j = sum of slices s = j (mod 4) ascending, in f64, rounded
// once to f32.
#include <type_traits>
#include "train_device.hpp"
#include "finalize_device.hpp"
#include "../../include/gymrl.h"

namespace {

using namespace gymrl;


using fin::kCUs;
// Column slices of the N = 256 weight-stationary kernels.  2: a [256 x 128] slice per workgroup (128 KiB of LDS, 8 row tiles
// per wave at 262,144 rows); 4: [256 x 64] slices (64 KiB, 16 row tiles per wave, 64 x 64 wave tiles) — the fixed cost of a
// launch (weight fill, first loads, the last tile's exposed epilogue) is halved and spread over twice the tiles, at the
// price of reading every activation row from L2 four times instead of twice.
constexpr int kSlices256 = 2;

enum { EPI_NONE = 0, EPI_TANH = 1, EPI_TANHBWD = 2, EPI_ADD = 3 };     // EPI_ADD: out = acc + H (H: a tensor of the output's shape)
constexpr bool epi_reads_h(int epi) { return epi == EPI_TANHBWD || epi == EPI_ADD; }

struct WsArgs {
  const float* A; int64_t M; int lda;
  const float* W; int ldw;
  float* out; int ldo;
  const float* bias;
  const float* H; int ldh;
  int slices;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- raw buffer access: out-of-range lanes read 0 / drop their store, so row tails need no masks or clamps,
// and the address is (descriptor base: SGPRs) + (lane offset: one VGPR) + immediate — no address VALU in the loops
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, uint32_t off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0);
}

__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, off, 0, 0);
}

// TRANS_W == false: W is [n][red] (torch Linear weight, forward);  true: W is [red][n] (input gradient).
//
// What the measurements on gfx950 say about an f32-MFMA kernel (tools/abl_gemm.py, profiles/r02_gemm_ablation.txt):
//  * v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and shares the SIMD's f32 lanes with the VALU: VALU
//    instructions do not hide in an MFMA's 64 cycles, neither from the same wave nor from a second wave on the SIMD
//    (every VALU instruction of the epilogue costs ~7 cycles of matrix time; two MFMA-bound waves per SIMD run no
//    faster than one).  So: ONE wave per SIMD with 128 accumulator registers, and an epilogue of as few VALU
//    instructions as the layer's arithmetic allows (tanh as train_tanhf's 1 - 2 / (exp2(x * 2 log2 e) + 1): 6 per element with the bias add);
//  * with the epilogue removed the loop below runs at 98 % of the MFMA peak (A straight from HBM, B from LDS);
//  * dword stores are issue-bound (4 per chunk cost 10 % of the kernel): the MFMA operands are swapped
//    (D = W_tile * X_tile^T) so that a lane holds 4 ADJACENT output columns of one row per accumulator quad and
//    stores / loads 16 bytes at a time.
// A wave owns 32*RT rows x BN columns (64 x 128 at RED = 256, 64 x 64 at RED = 512: the fully unrolled reduction
// loop must stay within hipcc's 16 K-instruction unroll budget and the instruction cache).  The epilogue of row tile t is spread over the MFMA stream of row tile t+1 —
// not for the VALU (see above) but so that its loads (H) and stores are never waited for: the finished accumulators
// move to `pend`; every chunk of the next tile carries a few epilogue instructions behind each MFMA, pinned by
// sched_barrier (left alone, hipcc emits them as one block in front of a vmcnt(0)).
template <int RED, int NT, int RT, bool TRANS_W, int EPI, int LDO, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_ws_kernel(WsArgs p) {
  constexpr int BN = 32 * NT, NCH = RED / 8, PF = 4, NGR = RT * NT * 4, CPG = NCH / NGR, REG = 4 * RT * NT, SPAN = REG * CPG;
  constexpr int kWaves = 4, kThreads = 256, ROWS = 32 * RT;
  constexpr int NST = EPI == EPI_TANH ? 6 : (EPI == EPI_TANHBWD ? 3 : 1);    // VALU stages per output element (EPI_ADD: 1)
  static_assert(NCH % PF == 0 && NCH % NGR == 0 && CPG >= 1 && 4 * NST <= SPAN - 1, "epilogue groups per chunk");
  static_assert(RED * BN * 4 <= 128 * 1024, "weight slice must fit LDS");
  __shared__ float lds[RED * BN + BN];                 // [q = 2c + h][n][4]: value W(red = 4q + e, n); then bias[BN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int S = p.slices, RG = gridDim.x / S, b = blockIdx.x;
  int slice, rg;
  if ((RG & 7) == 0) { const int xcd = b & 7, r = b >> 3; slice = r % S; rg = (r / S) * 8 + xcd; }
  else { slice = b % S; rg = b / S; }
  const int n0 = slice * BN;

  // ---- weight slice (+ bias slice) -> LDS, once per workgroup: all loads of a batch are issued before the first
  // LDS store (the fill is ~10 us of every launch; left rolled it is one L2 round trip per iteration)
  {
    constexpr int ITER = BN * (RED / 4) / kThreads, BATCH = 8;
    static_assert(BN * (RED / 4) % kThreads == 0 && ITER % BATCH == 0, "fill tiling");
#pragma unroll 1
    for (int it0 = 0; it0 < ITER; it0 += BATCH) {
      f32x4 v[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int idx = tid + (it0 + k) * kThreads, n = idx % BN, q = idx / BN;
        if constexpr (!TRANS_W) {
          v[k] = *reinterpret_cast<const f32x4*>(p.W + (size_t)(n0 + n) * p.ldw + 4 * q);
        } else {      // thread (q, n) gathers W[4q .. 4q+3][n0 + n]: four row reads, coalesced across the wave
          const float* w = p.W + (size_t)(4 * q) * p.ldw + n0 + n;
          v[k] = f32x4{w[0], w[p.ldw], w[2 * (size_t)p.ldw], w[3 * (size_t)p.ldw]};
        }
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int idx = tid + (it0 + k) * kThreads, n = idx % BN, q = idx / BN;
        *reinterpret_cast<f32x4*>(&lds[((size_t)q * BN + n) * 4]) = v[k];
      }
    }
  }
  if (tid < BN) lds[RED * BN + tid] = (!epi_reads_h(EPI) && p.bias) ? p.bias[n0 + tid] : 0.0f;
  __syncthreads();

  const int64_t M = p.M;
  const int64_t tasks = (M + ROWS - 1) / ROWS;
  const int64_t bt_count = (tasks + kWaves - 1) / kWaves;
  uint32_t aoff[RT];                                   // lane (i, h): row 32 rt + i of the tile, floats 4h..4h+3 of a chunk
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) aoff[rt] = (uint32_t)((32 * rt + i) * RED + 4 * h) * 4u;
  // D = W_tile X_tile^T: lane (i, h) holds row i of the row tile; accumulator quad g (registers 4g..4g+3) = columns
  // 8g + 4h .. 8g + 4h + 3 of the 32-column tile
  const uint32_t ooff = (uint32_t)(i * LDO + 4 * h) * 4u;
  const f32x4* bl = reinterpret_cast<const f32x4*>(lds) + (size_t)h * BN + i;   // + c * 2 * BN + 32 * nt
  const f32x4* bias4 = reinterpret_cast<const f32x4*>(lds + RED * BN) + h;      // + (32 nt + 8 g) / 4

  // descriptors over the valid rows of a row tile (0 rows for a tile past the end: loads give 0, stores vanish)
  auto rows_of = [&](int64_t task) {
    int64_t r = M - task * ROWS;
    return (uint32_t)(r < 0 ? 0 : (r > ROWS ? ROWS : r));
  };
  auto a_rsrc = [&](int64_t task) { return make_rsrc(p.A + task * ROWS * RED, rows_of(task) * RED * 4u); };
  auto o_rsrc = [&](int64_t task) {
    const uint32_t r = rows_of(task);
    return make_rsrc(p.out + task * ROWS * LDO + n0, r * LDO * 4u - (r ? n0 * 4u : 0u));
  };
  auto h_rsrc = [&](int64_t task) {
    const uint32_t r = (epi_reads_h(EPI) && p.H) ? rows_of(task) : 0u;
    return make_rsrc(p.H + task * ROWS * LDO + n0, r * LDO * 4u - (r ? n0 * 4u : 0u));
  };
  // group G = (rt, nt, g): 4 adjacent outputs of one row
  auto group_off = [&](int G) {
    const int rt = G / (NT * 4), nt = (G / 4) % NT, g = G % 4;
    return ooff + (uint32_t)((32 * rt * LDO + 32 * nt + 8 * g) * 4);
  };
  // stage k of one output element; x carries the element through the stages
  auto stage = [&](int k, float x, float v, float bv, float hval) {
    if constexpr (EPI == EPI_NONE) x = v + bv;
    if constexpr (EPI == EPI_TANH) {       // train_tanhf(v + bias), one instruction per stage
      if (k == 0) x = v + bv;
      if (k == 1) x = x * 2.885390081777927f;
      if (k == 2) x = __builtin_amdgcn_exp2f(x);
      if (k == 3) x = x + 1.0f;
      if (k == 4) x = __builtin_amdgcn_rcpf(x);
      if (k == 5) x = fmaf(-2.0f, x, 1.0f);
    }
    if constexpr (EPI == EPI_TANHBWD) {
      if (k == 0) x = hval * hval;
      if (k == 1) x = 1.0f - x;
      if (k == 2) x = p.H ? v * x : v;
    }
    if constexpr (EPI == EPI_ADD) x = v + hval;
    return x;
  };

  int64_t bt = rg;
  __amdgpu_buffer_rsrc_t cur = a_rsrc(bt * kWaves + wave);
  __amdgpu_buffer_rsrc_t po = make_rsrc(p.out, 0), ph = make_rsrc(p.out, 0);     // nothing pending yet
  f32x4 abuf[PF][RT];
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) abuf[u][rt] = bload4(cur, aoff[rt] + 32u * u);
  f32x16 acc[RT][NT], pend[RT][NT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) pend[rt][nt][r] = 0.0f;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 bv = zero4, xs = zero4;
  // H of epilogue group G lives in hring[G % HD]; it is loaded HD - 1 groups (>= 3 us of MFMAs) before its use —
  // one group ahead (0.7 us) was less than an HBM miss and stalled the dX kernels by 30 %
  constexpr int HD = 4;
  static_assert(NGR % HD == 0, "ring slots line up across row tiles");
  f32x4 hring[HD] = {zero4, zero4, zero4, zero4};
  f32x4 bc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bc[nt] = bl[32 * nt];

#pragma unroll 1
  for (; bt < bt_count; bt += RG) {
    const int64_t task = bt * kWaves + wave;
    const __amdgpu_buffer_rsrc_t nxt = a_rsrc((bt + RG) * kWaves + wave);
    const __amdgpu_buffer_rsrc_t co = o_rsrc(task), chh = h_rsrc(task);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int u = c % PF;
      const int G = c / CPG;                            // epilogue group of the previous row tile on this chunk
      f32x4 bn[NT];
      // A chunk = 4 k-pairs x RT*NT MFMAs = REG pinned regions of ONE MFMA.  Region R of a group's span also
      // carries stage R / 4 of the group's element R % 4, or one of the loads / the group's 16-byte store.
#pragma unroll
      for (int r = 0; r < REG; ++r) {
        const int e = r / (RT * NT), m = r % (RT * NT), rt = m / NT, nt = m % NT;
        const int R = (c % CPG) * REG + r;
        if (r == 0) {            // operands of the next chunk (chunk 0 of the next tile is the same LDS data)
#pragma unroll
          for (int n2 = 0; n2 < NT; ++n2) bn[n2] = (ABL & 2) ? bc[n2] : bl[(size_t)((c + 1) % NCH) * 2 * BN + 32 * n2];
        }
        if (R == 0 && !epi_reads_h(EPI)) bv = bias4[(32 * ((G / 4) % NT) + 8 * (G % 4)) / 4];
        if (epi_reads_h(EPI) && R == 8)             // H of group G + HD - 1 (past the tile's last group: this tile's own)
          hring[(G + HD - 1) % HD] = (G + HD - 1 < NGR) ? bload4(ph, group_off(G + HD - 1)) : bload4(chh, group_off(G + HD - 1 - NGR));
        if (c == 0 && e == 0) {
          f32x16 z;
#pragma unroll
          for (int q = 0; q < 16; ++q) z[q] = 0.0f;
          acc[rt][nt] = mfma32(bc[nt][e], abuf[u][rt][e], z);
        } else {
          acc[rt][nt] = mfma32(bc[nt][e], abuf[u][rt][e], acc[rt][nt]);
        }
        if (!(ABL & 4) && R < 4 * NST) {
          const int q = R % 4;
          xs[q] = stage(R / 4, xs[q], pend[G / (NT * 4)][(G / 4) % NT][4 * (G % 4) + q], bv[q], hring[G % HD][q]);
        }
        if (!(ABL & 4) && R == SPAN - 1) {
          if (ABL & 8) asm volatile("" :: "v"(xs)); else bstore4(po, group_off(G), xs);
        }
        if (r == REG - 1 && !(ABL & 1)) {     // refill the ring slot just consumed: chunk c + PF, or the next tile's head
#pragma unroll
          for (int r2 = 0; r2 < RT; ++r2)
            abuf[u][r2] = (c + PF < NCH) ? bload4(cur, aoff[r2] + 32u * (c + PF)) : bload4(nxt, aoff[r2] + 32u * (c + PF - NCH));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int n2 = 0; n2 < NT; ++n2) bc[n2] = bn[n2];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) pend[rt][nt] = acc[rt][nt];
    po = co;
    ph = chh;
    cur = nxt;
  }
  // the last row tile's epilogue has no MFMA stream to ride on
#pragma unroll
  for (int G = 0; G < NGR; ++G) {
    f32x4 hq = hring[G % HD], x4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (epi_reads_h(EPI) && G >= HD - 1) hq = bload4(ph, group_off(G));
    const f32x4 b4 = bias4[(32 * ((G / 4) % NT) + 8 * (G % 4)) / 4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < NST; ++k) x4[q] = stage(k, x4[q], pend[G / (NT * 4)][(G / 4) % NT][4 * (G % 4) + q], b4[q], hq[q]);
    bstore4(po, group_off(G), x4);
  }
}


// ------------------------------------------------------------------ narrow reductions (RED = 64 / 128) ------
// The 64- and 128-wide layers (PPO-full's mHC network at 262,144-row micro-batches: ppo_full_lunarlander.py:197-229, :287-318;
// ActorCritic at hidden_dim 64 / 128) are as much HBM- as MFMA-bound: 1 KB of operand + result per row against 33 kFLOP at
// 128 x 128.  Same product, same accumulation order as gemm_ws_kernel (so the oracle's fmaf chain restates both), but:
//   * the whole weight slice [RED x BN] is at most 64 KiB, which leaves LDS for the row tiles: a wave moves its next 32-row
//     tile from HBM with fully coalesced 16-byte loads (one contiguous 1-KiB segment per instruction: the tile is 32 * RED
//     consecutive floats) into registers while it multiplies the current one, then parks it in its own padded LDS tile
//     (pitch RED + 4 floats: the MFMA-layout read of 32 rows x 16 bytes is conflict free).  The direct-to-register
//     MFMA-layout loads of gemm_ws_kernel fetch 32 separate 32-byte pieces per instruction, which a 256 / 512-long reduction
//     amortises over its L1 hits and a 128-long one does not (the round-2 kernel without staging: 110 us against the
//     library's 97 at 262,144 x 128 x 128);
//   * nothing is shared between waves after the weight fill — no barrier inside the row loop, a wave's LDS traffic is
//     ordered by the LDS queue itself.
// A wave owns 32 rows x BN columns per tile; D = W_tile X_tile^T as above (lane (i, h) holds 4 adjacent output columns of
// row i per accumulator quad: 16-byte stores / H loads).
template <int RED, int NT, bool TRANS_W, int EPI, int LDO>
__global__ __launch_bounds__(256) void gemm_ns_kernel(WsArgs p) {
  constexpr int BN = 32 * NT, NCH = RED / 8, kWaves = 4, kThreads = 256, ROWS = 32;
  constexpr int APITCH = RED + 4, OPITCH = BN + 4;                  // padded pitches: conflict-free MFMA-layout accesses
  constexpr int TILE = ROWS * (APITCH > OPITCH ? APITCH : OPITCH);  // floats: the wave's tile holds A rows, then its output rows
  constexpr int NLD = ROWS * RED / 4 / 64, NST = ROWS * BN / 4 / 64; // float4 loads / stores per lane and row tile
  static_assert(RED % 8 == 0 && (ROWS * RED / 4) % 64 == 0 && (ROWS * BN / 4) % 64 == 0, "tile tiling");
  static_assert((RED * BN + BN + kWaves * TILE) * 4 <= 160 * 1024, "weight slice + row tiles must fit LDS");
  __shared__ float lds[RED * BN + BN];                 // [q = 2c + h][n][4]: value W(red = 4q + e, n); then bias[BN]
  __shared__ float tiles[kWaves][TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
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
  // the A tile is one contiguous block of 32 * RED floats: float4 number j * 64 + lane of it — a wave-wide load is one
  // contiguous 1-KiB segment
  auto fetch = [&](int64_t task, f32x4 (&v)[NLD]) {
    const __amdgpu_buffer_rsrc_t r = make_rsrc(p.A + task * ROWS * RED, rows_of(task) * RED * 4u);
#pragma unroll
    for (int j = 0; j < NLD; ++j) v[j] = bload4(r, (uint32_t)(j * 64 + lane) * 16u);
  };
  int64_t bt = rg;
  f32x4 nxt[NLD];
  fetch(bt * kWaves + wave, nxt);                       // in flight under the weight fill

  {
    constexpr int ITER = BN * (RED / 4) / kThreads, BATCH = ITER < 8 ? ITER : 8;
    static_assert(BN * (RED / 4) % kThreads == 0 && ITER % BATCH == 0, "fill tiling");
#pragma unroll 1
    for (int it0 = 0; it0 < ITER; it0 += BATCH) {
      f32x4 v[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int idx = tid + (it0 + k) * kThreads, n = idx % BN, q = idx / BN;
        if constexpr (!TRANS_W) {
          v[k] = *reinterpret_cast<const f32x4*>(p.W + (size_t)(n0 + n) * p.ldw + 4 * q);
        } else {
          const float* w = p.W + (size_t)(4 * q) * p.ldw + n0 + n;
          v[k] = f32x4{w[0], w[p.ldw], w[2 * (size_t)p.ldw], w[3 * (size_t)p.ldw]};
        }
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int idx = tid + (it0 + k) * kThreads, n = idx % BN, q = idx / BN;
        *reinterpret_cast<f32x4*>(&lds[((size_t)q * BN + n) * 4]) = v[k];
      }
    }
  }
  if (tid < BN) lds[RED * BN + tid] = (EPI != EPI_TANHBWD && p.bias) ? p.bias[n0 + tid] : 0.0f;
  __syncthreads();

  float* tile = tiles[wave];
  const f32x4* bl = reinterpret_cast<const f32x4*>(lds) + (size_t)h * BN + i;   // + c * 2 * BN + 32 * nt
  const f32x4* bias4 = reinterpret_cast<const f32x4*>(lds + RED * BN) + h;      // + (32 nt + 8 g) / 4
  const float* arow = tile + i * APITCH + 4 * h;                                // + 8 c
  float* orow = tile + i * OPITCH + 4 * h;                                      // + 32 nt + 8 g
  auto park = [&](const f32x4 (&v)[NLD]) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f4 = j * 64 + lane, r = f4 / (RED / 4), c4 = f4 % (RED / 4);
      *reinterpret_cast<f32x4*>(tile + r * APITCH + 4 * c4) = v[j];
    }
  };
  park(nxt);
#pragma unroll 1
  for (; bt < bt_count; bt += RG) {
    const int64_t task = bt * kWaves + wave;
    const uint32_t rows = rows_of(task);
    fetch((bt + RG) * kWaves + wave, nxt);             // past the end: zero rows, the loads return 0
    // H (tanh' factor) of this tile's output rows, coalesced like A: float4 j * 64 + lane of the [32][BN] block at column n0
    f32x4 hv[NST];
    if constexpr (EPI == EPI_TANHBWD) {
      const __amdgpu_buffer_rsrc_t chh = make_rsrc(p.H + task * ROWS * LDO + n0, p.H ? rows * LDO * 4u - (rows ? n0 * 4u : 0u) : 0u);
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        const int f4 = j * 64 + lane, r = f4 / (BN / 4), c4 = f4 % (BN / 4);
        hv[j] = bload4(chh, (uint32_t)(r * LDO + 4 * c4) * 4u);
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    // reduction: the next chunk's LDS operands are requested before the current chunk's 4 NT MFMAs (pinned: left alone, the
    // scheduler sinks every read next to its use behind an lgkmcnt(0))
    f32x4 a4 = *reinterpret_cast<const f32x4*>(arow), b4[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b4[nt] = bl[32 * nt];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      f32x4 an = a4, bn[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bn[nt] = b4[nt];
      if (c + 1 < NCH) {
        an = *reinterpret_cast<const f32x4*>(arow + 8 * (c + 1));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bn[nt] = bl[(size_t)(c + 1) * 2 * BN + 32 * nt];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(b4[nt][e], a4[e], acc[nt]);
      __builtin_amdgcn_sched_barrier(0);
      a4 = an;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[nt] = bn[nt];
    }
    // epilogue: the outputs go through the wave's tile (its A rows are consumed) so that the global stores are whole
    // contiguous row segments — lane (i, h) holds 4 adjacent columns of row i per accumulator quad
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = bias4[(32 * nt + 8 * g) / 4];
        f32x4 x4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = acc[nt][4 * g + q];
          if constexpr (EPI == EPI_TANH) {                 // train_tanhf(v + bias), as gemm_ws_kernel's stages
            float x = v + bv[q];
            x = x * 2.885390081777927f;
            x = __builtin_amdgcn_exp2f(x);
            x = x + 1.0f;
            x = __builtin_amdgcn_rcpf(x);
            x4[q] = fmaf(-2.0f, x, 1.0f);
          } else if constexpr (EPI == EPI_NONE) {
            x4[q] = v + bv[q];
          } else {
            x4[q] = v;                                     // the tanh' factor is applied on the way out (below)
          }
        }
        *reinterpret_cast<f32x4*>(orow + 32 * nt + 8 * g) = x4;
      }
    {
      const __amdgpu_buffer_rsrc_t co = make_rsrc(p.out + task * ROWS * LDO + n0, rows * LDO * 4u - (rows ? n0 * 4u : 0u));
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        const int f4 = j * 64 + lane, r = f4 / (BN / 4), c4 = f4 % (BN / 4);
        f32x4 x4 = *reinterpret_cast<const f32x4*>(tile + r * OPITCH + 4 * c4);
        if constexpr (EPI == EPI_TANHBWD) {
          if (p.H) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float x = hv[j][q] * hv[j][q];
              x = 1.0f - x;
              x4[q] = x4[q] * x;
            }
          }
        }
        bstore4(co, (uint32_t)(r * LDO + 4 * c4) * 4u, x4);
      }
    }
    park(nxt);                                          // (the LDS queue orders these writes behind this tile's reads)
  }
}

// ------------------------------------------------------------------ dW ------
struct TnArgs {
  const float* dY; int ldy;
  const float* X; int ldx;
  int64_t M, rows_per_slice;
  float* parts;            // f32[ntiles][slices][256][256]
  float* cs_parts;         // f32[ntiles][slices][256] column sums of dY per slice, or nullptr
  int slices, ntiles;
};

// LDY = row pitch of dY (256 or 512 floats); X has 256 columns.  Four waves, ONE per SIMD: two MFMA-bound waves
// sharing a SIMD's matrix pipe measured 70 TF/s (MFMAs only) against 145 for one wave with the same 256
// accumulator registers' worth of independent tiles, so a wave owns a 128 x 128 quarter of the output tile
// (4 x 4 tiles of 32 x 32 = 256 accumulator registers) and feeds 16 MFMAs from one dwordx4 of each operand.
// ABL (diagnostics, tools/abl_gemm.py): 0 product kernel; 1 no loads inside the loop; 2 MFMAs only.
constexpr int kTnThreads = 256;
template <int PF, int LDY, int ABL = 0>
__global__ __launch_bounds__(kTnThreads) void gemm_tn_kernel(TnArgs p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5, wn = wave >> 1, wk = wave & 1;
  const int S = p.slices, NTL = p.ntiles, b = blockIdx.x;
  int ntile, slice;
  if ((S & 7) == 0) { const int xcd = b & 7, r = b >> 3; ntile = r % NTL; slice = (r / NTL) * 8 + xcd; }
  else { ntile = b % NTL; slice = b / NTL; }
  const int64_t m0 = (int64_t)slice * p.rows_per_slice;
  const int64_t m1 = m0 + p.rows_per_slice < p.M ? m0 + p.rows_per_slice : p.M;
  const int steps = m1 > m0 ? (int)((m1 - m0 + 1) / 2) : 0;
  const uint32_t rows = m1 > m0 ? (uint32_t)(m1 - m0) : 0u;

  f32x16 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
  f32x4 cs = {0.0f, 0.0f, 0.0f, 0.0f};                 // column sums of dY over the rows this lane reads (wk == 0 waves)

  // lane (i, h) reads row m0 + 2 s + h of the slice: 4 adjacent columns of dY and of X.  The descriptors end with
  // the slice, so the odd row of an odd-sized slice and the ring slots past the last pair read as zeros.
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.dY + m0 * LDY, rows * LDY * 4u);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.X + m0 * 256, rows * 256 * 4u);
  uint32_t oa = (uint32_t)(h * LDY + ntile * 256 + 128 * wn + 4 * i) * 4u;
  uint32_t ob = (uint32_t)(h * 256 + 128 * wk + 4 * i) * 4u;
  if (steps > 0) {
    f32x4 abuf[PF], bbuf[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {     // same issue order as inside the loop: the compiler's loop-carried vmcnt counts stay exact
      abuf[u] = bload4(ra, oa);
      bbuf[u] = bload4(rb, ob);
      oa += 2 * LDY * 4;
      ob += 2 * 256 * 4;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const f32x4 a = abuf[u], bb = bbuf[u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            acc[t][v] = mfma32(a[t], bb[v], acc[t][v]);
            if (ABL == 0 && t == 0 && v == 1) {          // refill, issued behind the first MFMAs of the step
              abuf[u] = bload4(ra, oa);
              bbuf[u] = bload4(rb, ob);
              oa += 2 * LDY * 4;
              ob += 2 * 256 * 4;
            }
          }
        if (ABL < 2) cs += a;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // tile (t, v): D[i'][j'] = dW[n = 128 wn + 4 i' + t][k = 128 wk + 4 j' + v]
  float* out = p.parts + ((size_t)ntile * S + slice) * 65536;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ip = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int n = 128 * wn + 4 * ip + t, k = 128 * wk + 4 * i;
      *reinterpret_cast<f32x4*>(out + (size_t)n * 256 + k) = f32x4{acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
    }
  if (p.cs_parts && wk == 0) {          // even rows + odd rows of the slice, f32
#pragma unroll
    for (int t = 0; t < 4; ++t) cs[t] = cs[t] + __shfl_xor(cs[t], 32, 64);
    if (h == 0) *reinterpret_cast<f32x4*>(p.cs_parts + ((size_t)ntile * S + slice) * 256 + 128 * wn + 4 * i) = cs;
  }
}

// The slice partials' second halves: finalize_device.hpp (shared with gymrl_update_finalize's one launch)
__global__ __launch_bounds__(256) void tn_colsum_kernel(const float* __restrict__ parts, int slices, float* __restrict__ db) {
  __shared__ double sm[4][64];
  fin::tn_colsum_body(parts, slices, db, blockIdx.x, sm);
}
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ parts, int slices, int ldw,
                                                        float* __restrict__ dW) {
  __shared__ double sm[3][64][4];
  fin::tn_reduce_body(parts, slices, ldw, dW, blockIdx.x, sm);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Diagnostic knobs exist in the probe build only (make prof: -DGYMRL_PROF_BUILD, libgymrl_hip_prof.so, loaded by
// tools/abl_gemm.py through GYMRL_HIP_LIB).  The product library has no mutable global state and none of the
// timing-only kernel variants (they skip loads / epilogues and produce wrong results by design).
#ifdef GYMRL_PROF_BUILD
int g_tn_abl = 0;      // ablation mode of the weight-gradient kernel (0 = product kernel)
int g_ws_abl = 0;      // ablation mask of the forward kernel
int g_ws_slices256 = kSlices256;   // column slices of the 256-wide forward / input-gradient kernels (2 | 4): A/B of tools/micro_gemm.py
#else
constexpr int g_ws_slices256 = kSlices256;
#endif

// row groups of the weight-stationary kernels: at most one workgroup per CU (the weight slice fills LDS)
inline int ws_row_groups(int64_t M, int rows_per_task, int slices) {
  const int waves = 4;
  const int64_t tasks = (M + rows_per_task - 1) / rows_per_task;
  const int64_t bts = (tasks + waves - 1) / waves;
  int64_t rg = kCUs / slices;
  if (bts < rg) rg = bts;
  return (int)(rg < 1 ? 1 : rg);
}

template <int RED, int NT, int RT, bool TRANS_W, int EPI, int LDO>
void launch_ws(const WsArgs& a, int rg, hipStream_t s) {
  const dim3 grid(rg * a.slices), block(256);
#ifdef GYMRL_PROF_BUILD
  if constexpr (EPI == EPI_TANH && LDO == 256) {
    switch (g_ws_abl) {
      case 1: hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO, 1>), grid, block, 0, s, a); return;
      case 2: hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO, 2>), grid, block, 0, s, a); return;
      case 4: hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO, 4>), grid, block, 0, s, a); return;
      case 7: hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO, 7>), grid, block, 0, s, a); return;
      case 8: hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO, 8>), grid, block, 0, s, a); return;
      case 12: hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO, 12>), grid, block, 0, s, a); return;
      default: break;
    }
  }
#endif
  hipLaunchKernelGGL((gemm_ws_kernel<RED, NT, RT, TRANS_W, EPI, LDO>), grid, block, 0, s, a);
}

template <int RED, int NT, bool TRANS_W, int EPI, int LDO>
void launch_ns(const WsArgs& a, hipStream_t s) {
  // 32-row tasks; up to two workgroups per CU where the LDS footprint allows (the kernel is as much HBM- as MFMA-bound)
  constexpr int lds_bytes = (RED * 32 * NT + 32 * NT + 4 * 32 * (RED + 4)) * 4;
  const int per_cu = lds_bytes <= 80 * 1024 ? 2 : 1;
  const int64_t tasks = (a.M + 31) / 32, bts = (tasks + 3) / 4;
  int64_t rg = (int64_t)kCUs * per_cu / a.slices;
  if (bts < rg) rg = bts;
  if (rg < 1) rg = 1;
  hipLaunchKernelGGL((gemm_ns_kernel<RED, NT, TRANS_W, EPI, LDO>), dim3((unsigned)(rg * a.slices)), dim3(256), 0, s, a);
}

using fin::tn_geometry;
using fin::kColsumBytes;

}  // namespace

extern "C" {

#ifdef GYMRL_PROF_BUILD
int gymrl_gemm_config(int key, int value) {      // probe build only (include/gymrl.h)
  switch (key) {
    case 4: if (value < 0 || value > 2) return -22; g_tn_abl = value; return 0;
    case 5: if (value < 0 || value > 12) return -22; g_ws_abl = value; return 0;
    case 6: if (value != 2 && value != 4) return -22; g_ws_slices256 = value; return 0;
    default: return -22;
  }
}
#endif

size_t gymrl_gemm_workspace_bytes(void) {
  // column-sum partials of gymrl_linear_bwd_input + the partial tiles of gymrl_linear_bwd_weight
  // (ntiles * slices <= 256 tiles of 256 x 256 f32)
  return kColsumBytes + (size_t)kCUs * 65536 * sizeof(float) + 256;
}

int gymrl_linear_bwd_weight_geometry(int64_t B, int N, int* slices, int64_t* rows_per_slice) {
  if (B <= 0 || N < 256 || N % 256 || !slices || !rows_per_slice) return -22;
  tn_geometry(B, N, slices, rows_per_slice);
  return 0;
}

int gymrl_linear_fwd(const float* X, const float* W, const float* b, int64_t B, int K, int N, int act, float* Y,
                     void* stream) {
  const bool wide = K == 256 && (N == 256 || N == 512);
  const bool narrow = (K == 64 || K == 128) && (N == K || N == 2 * K);
  if (!X || !W || !Y || B < 0 || !(wide || narrow) || (act != 0 && act != 1) || !al16(X) || !al16(W) || !al16(Y))
    return -22;
  if (B == 0) return 0;
  WsArgs a{};
  a.A = X; a.M = B; a.lda = K; a.W = W; a.ldw = K; a.out = Y; a.ldo = N; a.bias = b;
  hipStream_t s = (hipStream_t)stream;
  if (wide) {
    a.slices = N / 128;
    const int rg = ws_row_groups(B, 64, a.slices);
    if (N == 256 && g_ws_slices256 == 4) {
      a.slices = 4;
      if (act == 1) launch_ws<256, 2, 2, false, EPI_TANH, 256>(a, ws_row_groups(B, 64, 4), s);
      else launch_ws<256, 2, 2, false, EPI_NONE, 256>(a, ws_row_groups(B, 64, 4), s);
    } else if (N == 256) {
      if (act == 1) launch_ws<256, 4, 2, false, EPI_TANH, 256>(a, rg, s);
      else launch_ws<256, 4, 2, false, EPI_NONE, 256>(a, rg, s);
    } else {
      if (act == 1) launch_ws<256, 4, 2, false, EPI_TANH, 512>(a, rg, s);
      else launch_ws<256, 4, 2, false, EPI_NONE, 512>(a, rg, s);
    }
  } else if (K == 128) {                      // column slices of 128
    a.slices = N / 128;
    if (N == 128) { if (act == 1) launch_ns<128, 4, false, EPI_TANH, 128>(a, s); else launch_ns<128, 4, false, EPI_NONE, 128>(a, s); }
    else { if (act == 1) launch_ns<128, 4, false, EPI_TANH, 256>(a, s); else launch_ns<128, 4, false, EPI_NONE, 256>(a, s); }
  } else {                                    // K == 64: column slices of 64
    a.slices = N / 64;
    if (N == 64) { if (act == 1) launch_ns<64, 2, false, EPI_TANH, 64>(a, s); else launch_ns<64, 2, false, EPI_NONE, 64>(a, s); }
    else { if (act == 1) launch_ns<64, 2, false, EPI_TANH, 128>(a, s); else launch_ns<64, 2, false, EPI_NONE, 128>(a, s); }
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_bwd_input(const float* dY, const float* W, const float* H, int64_t B, int N, int K, float* dX,
                           void* stream) {
  const bool wide = K == 256 && (N == 256 || N == 512);
  const bool narrow = (K == 64 || K == 128) && (N == K || N == 2 * K);
  if (!dY || !W || !dX || B < 0 || !(wide || narrow) || !al16(dY) || !al16(W) || !al16(dX) || (H && !al16(H)))
    return -22;
  if (B == 0) return 0;
  WsArgs a{};
  a.A = dY; a.M = B; a.lda = N; a.W = W; a.ldw = K; a.out = dX; a.ldo = K; a.H = H; a.ldh = K;
  hipStream_t s = (hipStream_t)stream;
  if (wide) {
    if (N == 256 && g_ws_slices256 == 4) {
      a.slices = 4;
      launch_ws<256, 2, 2, true, EPI_TANHBWD, 256>(a, ws_row_groups(B, 64, 4), s);
    } else if (N == 256) {
      a.slices = 2;
      launch_ws<256, 4, 2, true, EPI_TANHBWD, 256>(a, ws_row_groups(B, 64, 2), s);
    } else {
      a.slices = 4;
      launch_ws<512, 2, 2, true, EPI_TANHBWD, 256>(a, ws_row_groups(B, 64, 4), s);
    }
  } else if (K == 128) {
    a.slices = 1;                             // one 128-column slice: the whole [N x 128] matrix is the workgroup's
    if (N == 128) launch_ns<128, 4, true, EPI_TANHBWD, 128>(a, s);
    else launch_ws<256, 4, 2, true, EPI_TANHBWD, 128>(a, ws_row_groups(B, 64, 1), s);     // reduction over 256: streamed rows
  } else {
    a.slices = 1;
    if (N == 64) launch_ns<64, 2, true, EPI_TANHBWD, 64>(a, s);
    else launch_ns<128, 2, true, EPI_TANHBWD, 64>(a, s);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_bwd_input_add(const float* dY, const float* W, const float* G, int64_t B, int N, int K, float* dX, void* stream) {
  if (!dY || !W || !G || !dX || B < 0 || N != 256 || K != 128 || !al16(dY) || !al16(W) || !al16(dX) || !al16(G)) return -22;
  if (B == 0) return 0;
  WsArgs a{};
  a.A = dY; a.M = B; a.lda = N; a.W = W; a.ldw = K; a.out = dX; a.ldo = K; a.H = G; a.ldh = K;
  a.slices = 1;
  launch_ws<256, 4, 2, true, EPI_ADD, 128>(a, ws_row_groups(B, 64, 1), (hipStream_t)stream);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_linear_bwd_weight(const float* dY, const float* X, int64_t B, int N, int K, float* dW, float* db,
                            void* workspace, void* stream) {
  // dW == NULL: the slice partials only (they stay in `workspace`; db != NULL still asks for the column-sum partials) —
  // gymrl_update_finalize reduces them together with the minibatch's other reductions
  if (!dY || !X || !workspace || B <= 0 || K != 256 || N < 256 || N % 256 || N > 512 || !al16(dY) || !al16(X) ||
      (dW && !al16(dW)) || !al16(workspace))
    return -22;
  TnArgs a{};
  a.dY = dY; a.ldy = N; a.X = X; a.ldx = K; a.M = B; a.ntiles = N / 256;
  tn_geometry(B, N, &a.slices, &a.rows_per_slice);
  a.cs_parts = db ? (float*)workspace : nullptr;                    // first kColsumBytes of the workspace
  a.parts = (float*)((char*)workspace + kColsumBytes);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a.slices * a.ntiles), block(kTnThreads);
#ifdef GYMRL_PROF_BUILD
  if (g_tn_abl) {        // timing-only variants (wrong results by design)
    if (g_tn_abl == 1) hipLaunchKernelGGL((gemm_tn_kernel<8, 256, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_tn_kernel<8, 256, 2>), grid, block, 0, s, a);
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
#endif
  if (N == 256) hipLaunchKernelGGL((gemm_tn_kernel<8, 256>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gemm_tn_kernel<8, 512>), grid, block, 0, s, a);
  if (dW) {
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(a.ntiles * 256), dim3(256), 0, s, a.parts, a.slices, K, dW);
    if (db) hipLaunchKernelGGL(tn_colsum_kernel, dim3(a.ntiles * 4), dim3(256), 0, s, a.cs_parts, a.slices, db);
  }
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
