// mlp_device.hpp — device side of the one-launch MLP forward (see mlp.hip for the design notes):
// shared by mlp_forward_kernel and the persistent rollout kernel.
#pragma once
#include "gymrl_device.hpp"
#include "../../include/gymrl.h"

namespace gymrl {
namespace mlp {

constexpr int kRows   = 16;                 // batch rows per workgroup = one MFMA M tile
constexpr int kMaxW   = GYMRL_MLP_MAX_WIDTH; // widest layer input/output held in LDS
constexpr int kStride = kMaxW + 4;          // +4 floats: rows start on different banks
constexpr int kBufs   = 3;                  // ping-pong activation buffers
constexpr int kMaxIn  = GYMRL_MLP_MAX_INPUT; // widest network input (observation)
constexpr int kInStride = kMaxIn + 4;   // kMaxIn is a multiple of 64
constexpr int kWaves  = 4;
constexpr int kHeadStride = 16;             // persistent rollout: [16 rows][<= 16 output columns] LDS tile
constexpr int kLdsBytes = 96 * 1024;        // > 80 KB: one workgroup per CU
constexpr int kTPW    = 4;                  // 16-column tiles a wave accumulates together

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Probe build (make prof): 100 MHz ticks workgroup 0's first lane spends in every stage interval of forward_tile, summed over
// the calls ([15] counts them); read back by the translation unit that wants them (rollout_lunar.hip: gymrl_mlp_fwd_prof_read).
#ifdef GYMRL_LUNAR_PROF
static __device__ unsigned long long g_fwd_prof[16];
#define MLP_FWD_MARK(i) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = wall_clock64(); g_fwd_prof[i] += now_ - fwd_t_; fwd_t_ = now_; } } while (0)
#else
#define MLP_FWD_MARK(i) do {} while (0)
#endif

__device__ __forceinline__ float activate(float x, int act) {
#ifdef GYMRL_ABL_NOTANH               // (timing ablation of the probe tools: wrong values)
  if (act == GYMRL_ACT_TANH) return x * 0.5f;
#endif
  if (act == GYMRL_ACT_TANH) return det_tanhf_sel(x);
  if (act == GYMRL_ACT_RELU) return fmaxf(x, 0.0f);
  return x;
}

// Register batch of weights: UK K-blocks (16 k each) x NT tiles, one float4 per lane.
template <int NT, int UK>
struct Batch { f32x4 w[UK][NT]; };

// Weights are read from the PACKED layout gymrl_mlp_pack writes (once per parameter update):
//   P[tile][kblock][lane][c] = W[16*tile + (lane&15)][16*kblock + 4*(lane>>4) + c]
// zero-padded to whole tiles and to K rounded up to 64, i.e. exactly the MFMA B-operand
// register image.  A wave-wide dwordx4 load is then ONE contiguous 1-KiB segment (8 full
// cache lines, consecutive loads walk consecutive L2 channels) and needs no bounds checks.
// Reading torch's [out][in] rows directly costs 16 half-lines per load whose 1-KiB row
// stride lands on 4 of the 16 L2 channels: 8.4 us per 256x256 layer instead of < 2.
template <int NT, int UK>
__device__ __forceinline__ void load_batch(Batch<NT, UK>& b, const float* __restrict__ P, int nkb, int kb0, int t0,
                                           int lane) {
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const f32x4* tile = reinterpret_cast<const f32x4*>(P + ((size_t)(t0 + i) * nkb + kb0) * 256) + lane;
#pragma unroll
    for (int u = 0; u < UK; ++u) b.w[u][i] = tile[64 * u];
  }
}

template <int NT, int UK>
__device__ __forceinline__ void mfma_batch(f32x4 (&acc)[NT], const Batch<NT, UK>& b, const float* __restrict__ A,
                                           int a_stride, int kb0, int lane) {
  const float* ap = A + (lane & 15) * a_stride + 16 * kb0 + 4 * (lane >> 4);
#pragma unroll
  for (int u = 0; u < UK; ++u) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 16 * u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#ifdef GYMRL_ABL_NOMFMA               // (timing ablation: the operands stay live, the matrix pipe idle)
        acc[i][j] += a[j] * b.w[u][i][j];
#else
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b.w[u][i][j], acc[i], 0, 0, 0);
#endif
    }
  }
}

// NT adjacent 16-column tiles of one stage, K = NB batches of UK K-blocks: straight-line code, the
// next batch's loads are issued before the current batch's MFMAs, so the compiler's vmcnt waits
// leave exactly one batch in flight behind the matrix pipe.
template <int NT, int UK, int NB>
__device__ __forceinline__ void run_tiles_k(const gymrl_mlp_stage& st, const float* __restrict__ A, int a_stride,
                                            float* __restrict__ dst_lds, float* __restrict__ head_lds, int head_col,
                                            int t0, int m0, int n_rows, int lane) {
  const int nkb = ((st.in_dim + 63) & ~63) >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  Batch<NT, UK> b[2];
  load_batch<NT, UK>(b[0], st.W, nkb, 0, t0, lane);
  float bias[NT];                      // issued now, consumed in the epilogue
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = min(16 * (t0 + i) + (lane & 15), st.out_dim - 1);
    bias[i] = st.b ? st.b[col] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (i + 1 < NB) load_batch<NT, UK>(b[(i + 1) & 1], st.W, nkb, (i + 1) * UK, t0, lane);
    // keep the machine scheduler from sinking the prefetch down to its first use
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch<NT, UK>(acc, b[i & 1], A, a_stride, i * UK, lane);
    __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue: D[row = 4*(lane>>4)+r][col = lane&15] + bias, activation, to LDS or HBM
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = 16 * (t0 + i) + (lane & 15);
    if (col < st.out_dim) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        const float v = activate(acc[i][r] + bias[i], st.act);
        if (dst_lds) dst_lds[row * kStride + col] = v;
        else if (head_lds) head_lds[row * kHeadStride + head_col + col] = v;     // persistent rollout: outputs stay on the CU
        else if (m0 + row < n_rows) st.out[(size_t)(m0 + row) * st.out_stride + col] = v;
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void run_tiles(const gymrl_mlp_stage& st, const float* A, int a_stride, float* dst_lds,
                                          float* head_lds, int head_col, int t0, int m0, int n_rows, int lane) {
  const int nb64 = (st.in_dim + 63) >> 6;      // batches of 4 K-blocks; <= 4 for widths <= 256
#define GYMRL_MLP_CASE(UK_, NB_) run_tiles_k<NT, UK_, NB_>(st, A, a_stride, dst_lds, head_lds, head_col, t0, m0, n_rows, lane)
  // up to 16 float4 weight loads per lane in flight per batch
  if (nb64 == 4)      { if constexpr (NT == 1) GYMRL_MLP_CASE(16, 1); else if constexpr (NT == 2) GYMRL_MLP_CASE(8, 2); else GYMRL_MLP_CASE(4, 4); }
  else if (nb64 == 2) { if constexpr (NT <= 2) GYMRL_MLP_CASE(8, 1); else GYMRL_MLP_CASE(4, 2); }
  else if (nb64 == 1) GYMRL_MLP_CASE(4, 1);
  else GYMRL_MLP_CASE(4, 3);
#undef GYMRL_MLP_CASE
}

__device__ __forceinline__ void run_stage(const gymrl_mlp_stage& st, const float* A, int a_stride, float* dst_lds,
                                          float* head_lds, int head_col, int m0, int n_rows, int lane, int wave,
                                          int nwaves = kWaves) {
  const int ntiles = (st.out_dim + 15) >> 4;
  const int per_wave = (ntiles + nwaves - 1) / nwaves;
  for (int tb = 0; tb < per_wave; tb += kTPW) {              // one pass for widths <= 256
    const int t0 = wave * per_wave + tb;
    const int ntile = min(kTPW, min(per_wave - tb, ntiles - t0));
    if (ntile >= 4) run_tiles<4>(st, A, a_stride, dst_lds, head_lds, head_col, t0, m0, n_rows, lane);
    else if (ntile == 3) run_tiles<3>(st, A, a_stride, dst_lds, head_lds, head_col, t0, m0, n_rows, lane);
    else if (ntile == 2) run_tiles<2>(st, A, a_stride, dst_lds, head_lds, head_col, t0, m0, n_rows, lane);
    else if (ntile == 1) run_tiles<1>(st, A, a_stride, dst_lds, head_lds, head_col, t0, m0, n_rows, lane);
  }
}

// Two consecutive stages that touch disjoint buffers (e.g. the actor's and the critic's hidden layer, both reading
// the shared trunk; or the two heads) need no barrier between them.
__device__ __forceinline__ bool stages_independent(const gymrl_mlp_stage& a, const gymrl_mlp_stage& b) {
  const bool b_reads_a = a.dst >= 0 && b.src == a.dst;
  const bool b_overwrites_a_input = b.dst >= 0 && b.dst == a.src;
  const bool same_dst = a.dst >= 0 && b.dst == a.dst;
  return !b_reads_a && !b_overwrites_a_input && !same_dst;
}

// One whole network on the 16 rows whose input tile is already in `xin` (zero padded to 64 columns):
// stages ping-pong through `lds`; dst == -1 stages go to HBM (head_lds == nullptr) or to the head tile.
// Stages whose bit is set in `skip` are left out (their head columns stay reserved): forward_deferred runs them later.
__device__ __forceinline__ void forward_tile(const gymrl_mlp_desc& d, float (*lds)[kRows * kStride], float* xin,
                                             float* head_lds, int m0, int n_rows, int tid, uint32_t skip = 0u) {
  const int lane = tid & 63, wave = tid >> 6;
  int head_col = 0;
  bool paired = false;                  // this stage runs in the previous stage's barrier interval
#ifdef GYMRL_LUNAR_PROF
  unsigned long long fwd_t_ = wall_clock64();
  if (blockIdx.x == 0 && tid == 0) g_fwd_prof[15] += 1;
#endif
  for (int s = 0; s < d.n_stages; ++s) {
    const gymrl_mlp_stage st = d.stage[s];
    if ((skip >> s) & 1u) { if (st.dst < 0) head_col += st.out_dim; continue; }
    const float* A = st.src < 0 ? xin : lds[st.src];
    const int a_stride = st.src < 0 ? kInStride : kStride;
    float* dst_lds = st.dst >= 0 ? lds[st.dst] : nullptr;
    // the second stage of a pair starts its tile assignment one wave further on, so two single-tile stages
    // (the heads) run side by side on two SIMDs instead of back to back on one
    run_stage(st, A, a_stride, dst_lds, head_lds, head_col, m0, n_rows, lane, paired ? (wave + kWaves - 1) % kWaves : wave);
    if (st.dst < 0) head_col += st.out_dim;
    // columns between out_dim and the next multiple of 64 must read as zero: they are the next stage's K padding
    if (st.dst >= 0 && (st.out_dim & 63)) {
      const int lo = st.out_dim, hi = (st.out_dim + 63) & ~63;
      for (int e = tid; e < kRows * (hi - lo); e += kWaves * 64) {
        const int r = e / (hi - lo), c = lo + e % (hi - lo);
        lds[st.dst][r * kStride + c] = 0.0f;
      }
    }
    int nx = s + 1;                      // the next stage that runs here
    while (nx < d.n_stages && ((skip >> nx) & 1u)) ++nx;
    const bool pair_next = !paired && nx < d.n_stages && stages_independent(st, d.stage[nx]);
    if (!pair_next) __syncthreads();
    paired = pair_next;
    MLP_FWD_MARK(s);
  }
}

// Which stages only feed head columns >= first_col (for an actor-critic: the critic's layers and the value head)?  They can
// run after the others — on another wave, beside whatever consumes the first head columns — if they do not read the network
// input (rewritten meanwhile) and no stage that stays behind writes a buffer they read or write.  0 = nothing to defer.
__device__ __forceinline__ uint32_t deferrable_stages(const gymrl_mlp_desc& d, int first_col) {
  uint32_t m = 0u;
  int col[GYMRL_MLP_MAX_STAGES], hc = 0;
  for (int s = 0; s < d.n_stages; ++s) { col[s] = hc; if (d.stage[s].dst < 0) hc += d.stage[s].out_dim; }
  for (int s = d.n_stages - 1; s >= 0; --s) {
    const gymrl_mlp_stage& st = d.stage[s];
    if (st.dst < 0) { if (col[s] >= first_col) m |= 1u << s; continue; }
    bool any = false, all = true;
    for (int r = s + 1; r < d.n_stages; ++r) {
      if (d.stage[r].src == st.dst) { any = true; all = all && ((m >> r) & 1u); }
      if (d.stage[r].dst == st.dst) break;                 // the buffer is rewritten: later readers see another value
    }
    if (any && all) m |= 1u << s;
  }
  bool ok = m != 0u && m != (1u << d.n_stages) - 1u;
  for (int s = 0; s < d.n_stages && ok; ++s) {
    if (!((m >> s) & 1u)) continue;
    const gymrl_mlp_stage& st = d.stage[s];
    if (st.src < 0 || ((st.dst >= 0) && (st.out_dim & 63))) ok = false;
    for (int r = s + 1; r < d.n_stages && ok; ++r)
      if (!((m >> r) & 1u) && d.stage[r].dst >= 0 && (d.stage[r].dst == st.src || d.stage[r].dst == st.dst)) ok = false;
  }
  return ok ? m : 0u;
}

// The stages forward_tile(..., skip) left out, by ONE wave, in order (no workgroup barrier: the wave's own LDS writes and
// reads are in order).
__device__ __forceinline__ void forward_deferred(const gymrl_mlp_desc& d, float (*lds)[kRows * kStride], float* head_lds,
                                                 int m0, int n_rows, int lane, uint32_t skip) {
  int head_col = 0;
  for (int s = 0; s < d.n_stages; ++s) {
    const gymrl_mlp_stage st = d.stage[s];
    if ((skip >> s) & 1u) {
      run_stage(st, lds[st.src], kStride, st.dst >= 0 ? lds[st.dst] : nullptr, head_lds, head_col, m0, n_rows, lane, 0, 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (st.dst < 0) head_col += st.out_dim;
  }
}

}  // namespace mlp
}  // namespace gymrl
