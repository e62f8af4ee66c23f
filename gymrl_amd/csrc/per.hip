// per.hip — prioritised replay sum-tree on the GPU (float64, reference op order).
//
//   S1  SumTree.update / get_index / priority_max   rainbow_dqn_cartpole.py:116-152
//   S3  PrioritizedNStepBuffer.sample               :220-256
//   S4  update_priorities                           :258-261
//   S1'-S4' variant B                               ddqn_per_cartpole.py:67-147
//
// The reference's tree is updated incrementally (every ancestor += p - old leaf), one
// element after another, so its float64 contents depend on the ORDER of additions.  The
// batched update keeps that order per node: pass 1 resolves duplicate leaves (a later
// element sees the earlier element's priority as "old") and writes leaves last-writer-
// wins; pass 2 gives one workgroup to each tree depth, where every node's additions are
// applied by the first batch element that touches it, walking the batch in order.
// Nodes are independent of each other, so the result equals the sequential loop bit for
// bit.  That is update_priorities (S4), whose order the reference defines.  The N-row STORE of a vector step
// (idx == NULL) has no reference order — the reference stores one row per step — and is defined as one pairwise-summed
// addition per ancestor (per_store_device.hpp; identical to the reference at N = 1): per_store_* below.
// Sampling is a pointer chase (ceil(log2 cap)+1 dependent f64 loads per draw):
// latency-bound, one lane per draw, upper tree levels stay L2-resident.
#include <rocprim/block/block_radix_sort.hpp>
#include "gymrl_device.hpp"
#include "per_store_device.hpp"
#include "../../include/gymrl.h"

using namespace gymrl;

namespace {

constexpr int kBlock = 256;
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ int depth_of(int64_t t) { return 63 - __clzll((unsigned long long)(t + 1)); }

// workspace layout: [leaf i64[B]] [change f64[B]] [scratch f64[...]]
struct Ws {
  int64_t* leaf; double* change; double* partial;
  __host__ __device__ Ws(void* p, int B) {
    char* c = (char*)p;
    leaf = (int64_t*)c; c += sizeof(int64_t) * (size_t)B;
    change = (double*)c; c += sizeof(double) * (size_t)B;
    partial = (double*)c;
  }
};

__device__ __forceinline__ double prio_of(const double* prio, const double* ps_dev, double ps, int i) {
  return prio ? prio[i] : (ps_dev ? ps_dev[0] : ps);
}

// The batch's leaf ids and changes are staged in LDS (B <= kLdsB) so that the ordered searches and the
// per-node sequential sums below walk LDS, not HBM: a node's additions are one dependent chain of float64
// adds by construction (the reference's order), and with a global load per link the root of a
// 8192-row store cost 0.9 ms; from LDS it is the add latency (~35 us).
constexpr int kLdsB = 8192;

// pass 1 (single workgroup): leaves + per-element change of an explicit index batch (update_priorities).
__global__ __launch_bounds__(1024) void per_leaf_kernel(double* __restrict__ tree, int64_t cap,
                                                        const int32_t* __restrict__ idx, int idx_is_tree,
                                                        const double* __restrict__ prio,
                                                        const double* __restrict__ ps_dev, double ps,
                                                        int B, int64_t* __restrict__ leaf_out,
                                                        double* __restrict__ change_out, int use_lds) {
  extern __shared__ int64_t s_leaf_dyn[];
  const int64_t* lf = use_lds ? s_leaf_dyn : leaf_out;
  // phase A: leaf index of every element
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int64_t leaf = idx_is_tree ? (int64_t)idx[i] : (int64_t)idx[i] + cap - 1;
    leaf_out[i] = leaf;
    if (use_lds) s_leaf_dyn[i] = leaf;
  }
  __syncthreads();
  // phase B: change_i = p_i - (value of the leaf just before element i is applied)
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int64_t leaf = lf[i];
    double prev = tree[leaf];
    // duplicates possible: the latest earlier element on the same leaf
    // (eight LDS reads in flight per round: one at a time, each link of the search is a dependent ~100-clock round trip)
    int found = -1;
    for (int j0 = i - 1; j0 >= 0 && found < 0; j0 -= 8) {
      int64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j0 - u >= 0 ? lf[j0 - u] : (int64_t)-1;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (found < 0 && v[u] == leaf) found = j0 - u;
    }
    if (found >= 0) prev = prio_of(prio, ps_dev, ps, found);
    change_out[i] = prio_of(prio, ps_dev, ps, i) - prev;
  }
  __syncthreads();
  // phase C: last writer wins
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int64_t leaf = lf[i];
    bool last = true;
    for (int j0 = i + 1; j0 < B && last; j0 += 8) {
      int64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j0 + u < B ? lf[j0 + u] : (int64_t)-1;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (v[u] == leaf) last = false;
    }
    if (last) tree[leaf] = prio_of(prio, ps_dev, ps, i);
  }
}

// pass 2: blockIdx.x = node depth d (0 = root).  A node's additions happen in batch order (the reference's loop).
__global__ __launch_bounds__(1024) void per_ancestor_kernel(double* __restrict__ tree,
                                                            const int64_t* __restrict__ leaf_g,
                                                            const double* __restrict__ change_g, int B,
                                                            int use_lds) {
  extern __shared__ __attribute__((aligned(16))) int64_t s_dyn[];
  const int d = blockIdx.x;
  const int64_t* leaf = leaf_g;
  const double* change = change_g;
  if (use_lds) {
    int64_t* sl = s_dyn;
    double* sc = reinterpret_cast<double*>(s_dyn + B);
    for (int i = threadIdx.x; i < B; i += blockDim.x) { sl[i] = leaf_g[i]; sc[i] = change_g[i]; }
    __syncthreads();
    leaf = sl; change = sc;
  }
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int64_t lf = leaf[i];
    const int L = depth_of(lf);
    if (L <= d) continue;                                  // this element has no ancestor at depth d
    const int64_t node = ((lf + 1) >> (L - d)) - 1;
    // leader = first batch element reaching this node
    bool leader = true;
    for (int j0 = 0; j0 < i && leader; j0 += 8) {          // eight reads in flight per round (see per_leaf_kernel)
      int64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j0 + u < i ? leaf[j0 + u] : (int64_t)0;      // 0 = the root: never has an ancestor
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int Lj = depth_of(v[u]);
        if (Lj > d && (((v[u] + 1) >> (Lj - d)) - 1) == node) leader = false;
      }
    }
    if (!leader) continue;
    double acc = tree[node];
    for (int j0 = i; j0 < B; j0 += 8) {                    // the node's additions in batch order, operands fetched eight at a time
      int64_t v[8]; double c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const bool in = j0 + u < B; v[u] = in ? leaf[j0 + u] : (int64_t)0; c[u] = in ? change[j0 + u] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int Lj = depth_of(v[u]);
        if (Lj > d && (((v[u] + 1) >> (Lj - d)) - 1) == node) acc += c[u];
      }
    }
    tree[node] = acc;
  }
}

// ---- explicit index batches of up to kSmallB elements (update_priorities at the reference's batch sizes) ------------------
// Rounds 2-4 resolved "who touched this leaf / node before me" with per-element SEARCHES through LDS (dependent ~100-clock
// round trips, O(B) each: 15 + 28 us for 256 priorities in two launches, a quarter of a Rainbow vector step).  Here every
// element is a LANE and the batch walks past it: a wave holds 64 candidates, reads the ids of the others one at a time as a
// scalar (v_readlane of a register that holds 64 of them) and counts with vector compares — matches before me (my place in
// my node's run), matches in all (the run's length), the first match (the run's leader).  ~5 vector instructions per pair
// of 64 candidates, four SIMDs per compute unit, the batch's range split over the waves and combined with LDS atomics (integer:
// order-free).  A node's elements are then staged contiguously, in batch order, and its leader adds them one after another —
// the one thing the reference's loop (:258-261 -> :122-128) makes sequential.  Same float64 sums in the same order: the tree
// stays bit-identical (tests/test_hip_parity_offpolicy.py::test_sumtree_*).  (A first version of this round asked the same
// questions with wave-wide ballots and bit scans of the masks in SCALAR registers: 150 scalar instructions per element and ONE
// scalar unit per compute unit for its 16 waves — 28 us; this form: tools/micro_per.py.)
//
// ONE launch (gymrl_per_update_td with a ticket): blocks 0 .. depth-1 take one tree depth each — every one of them derives
// leaves, priorities and the per-element change itself from (idx, td, the OLD leaf values), nothing is handed over —, blocks
// depth .. depth+nb-1 scan the leaves for the new maximum (the batch's leaves masked out by a bitmap and replaced by their
// final values), and the LAST block to finish (a ticket) writes the leaves and folds the maxima: every reader of an old
// leaf value has then passed the ticket.  Without a ticket (gymrl_per_update) the same code runs as two launches: a leaf
// block that also publishes the changes, then the depth blocks.
constexpr int kSmallB = 512;
constexpr int kSmallWaves = 16;                // 1024 threads
constexpr int kMaxChunk = 8192;                // leaves per maximum block (8 per thread)
constexpr int32_t kNoFirst = 0x7fffffff;

// |td| -> priority, update_priorities' transform (rainbow_dqn_cartpole.py:258-261): float32 arithmetic like the reference —
// np.abs(f32) + python float and ** python float stay float32 under NumPy >= 2 —, x**a as exp(a log x) with the reproducible
// f32 kernels
__device__ __forceinline__ double td_priority(float td, double alpha, double eps, double clip) {
  float e = fabsf(td) + (float)eps;
  if (clip > 0.0 && e > (float)clip) e = (float)clip;
  return (double)det_expf((float)alpha * det_logf(e));
}

struct TdPrio { const float* td; double alpha, eps, clip; };   // td != nullptr: the batch's priorities are td_priority(td[i])
struct MaxLeaf { int64_t cap; double* partial; double* out; unsigned int* ticket; int nb; };
enum { SMALL_FUSED = 0, SMALL_LEAF = 1, SMALL_ANC = 2 };
struct SmallArgs {
  double* tree; int64_t cap; const int32_t* idx; int idx_is_tree; const double* prio; const double* ps_dev; double ps; int B;
  TdPrio tp; int depth; MaxLeaf mx; double* change_ws; int mode;
};
struct SmallLds {
  int32_t leaf[kSmallB];
  double p[kSmallB], old[kSmallB], change[kSmallB], stage[kSmallB];
  uint8_t last[kSmallB];
  int32_t prev[kSmallB], lastm[kSmallB];                       // duplicates: the latest earlier / the latest element on my leaf
  int32_t node[kSmallB], rank[kSmallB], cnt[kSmallB], first[kSmallB], slot[kSmallB];
};

// Wave roles: candidate group g (elements 64 g + lane) x part q of the batch's range.
struct Roles { int g, q, j0, j1, i; bool valid; };
__device__ __forceinline__ Roles roles(int B, int wave, int lane) {
  int groups = 1;
  while (groups * 64 < B) groups <<= 1;                            // 1, 2, 4, 8
  const int parts = kSmallWaves / groups, per = (B + parts - 1) / parts;
  Roles r;
  r.g = wave & (groups - 1); r.q = wave / groups;
  r.j0 = r.q * per; r.j1 = r.j0 + per < B ? r.j0 + per : B;
  r.i = 64 * r.g + lane; r.valid = r.i < B;
  return r;
}

__device__ __forceinline__ int small_node(int32_t leaf, int d) {          // the leaf's ancestor at depth d, -1: none
  const int L = 31 - __clz(leaf + 1);
  return L > d ? ((leaf + 1) >> (L - d)) - 1 : -1;
}

// batch -> LDS: leaves, new priorities and (want_old) the leaves' current values
__device__ __forceinline__ void small_load(const SmallArgs& a, SmallLds& s, bool want_old) {
  for (int i = threadIdx.x; i < a.B; i += blockDim.x) {
    const int32_t leaf = a.idx_is_tree ? a.idx[i] : a.idx[i] + (int32_t)(a.cap - 1);
    s.leaf[i] = leaf;
    if (want_old) s.old[i] = a.tree[leaf];
    s.p[i] = a.tp.td ? td_priority(a.tp.td[i], a.tp.alpha, a.tp.eps, a.tp.clip) : prio_of(a.prio, a.ps_dev, a.ps, i);
    s.prev[i] = -1; s.lastm[i] = -1;
  }
}

// change_i = p_i - (value of the leaf just before element i is applied: the latest earlier element on the same leaf, else the
// tree); last_i = no later element writes the same leaf (want_change false: last_i alone — the maximum blocks, which never
// read an old leaf value).  Ends with the results visible to the block.
__device__ __forceinline__ void small_dups(const SmallArgs& a, SmallLds& s, int wave, int lane, bool want_change) {
  const Roles r = roles(a.B, wave, lane);
  const int32_t mine = r.valid ? s.leaf[r.i] : -2;
  int prev = -1, lastm = -1;
  for (int jb = r.j0; jb < r.j1; jb += 64) {
    const int32_t vj = jb + lane < r.j1 ? s.leaf[jb + lane] : -1;
#pragma unroll
    for (int u = 0; u < 64; ++u) {                                 // (entries beyond the range hold a value that matches nothing)
      const int32_t L = __builtin_amdgcn_readlane(vj, u);
      const int j = jb + u;
      const bool eq = L == mine;
      lastm = eq ? j : lastm;                                      // (ascending: ends as the latest)
      prev = (eq && j < r.i) ? j : prev;
    }
  }
  if (r.valid) {
    if (prev >= 0) atomicMax(&s.prev[r.i], prev);
    if (lastm >= 0) atomicMax(&s.lastm[r.i], lastm);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.B; i += blockDim.x) {
    if (want_change) s.change[i] = s.p[i] - (s.prev[i] >= 0 ? s.p[s.prev[i]] : s.old[i]);
    s.last[i] = s.lastm[i] == i ? 1 : 0;                           // (i matches itself: lastm >= i)
  }
  __syncthreads();
}

// The additions of depth d's nodes, each node's in batch order.
__device__ __forceinline__ void small_ancestors(const SmallArgs& a, SmallLds& s, int d, int wave, int lane) {
  const Roles r = roles(a.B, wave, lane);
  double treeval = 0.0;
  for (int i = threadIdx.x; i < a.B; i += blockDim.x) {           // (B <= 512 < 1024 threads: one element per thread, i == r.i of part 0)
    const int32_t n = small_node(s.leaf[i], d);
    s.node[i] = n; s.rank[i] = 0; s.cnt[i] = 0; s.first[i] = kNoFirst; s.slot[i] = 0;
    if (n >= 0) treeval = a.tree[n];                               // requested now, needed by the node's leader at the end
  }
  __syncthreads();
  const int32_t mine = r.valid ? s.node[r.i] : -1;
  const int32_t key = mine >= 0 ? mine : -3;                       // (no ancestor at this depth: matches nothing)
  {
    int rank = 0, cnt = 0, first = kNoFirst;
    for (int jb = r.j0; jb < r.j1; jb += 64) {
      const int32_t vj = jb + lane < r.j1 ? s.node[jb + lane] : -1;
    #pragma unroll
    for (int u = 0; u < 64; ++u) {                                 // (entries beyond the range hold a value that matches nothing)
        const int32_t L = __builtin_amdgcn_readlane(vj, u);
        const int j = jb + u;
        const bool eq = L == key;
        cnt += eq ? 1 : 0;
        rank += (eq && j < r.i) ? 1 : 0;
        first = min(first, eq ? j : kNoFirst);
      }
    }
    if (mine >= 0 && cnt > 0) {
      atomicAdd(&s.cnt[r.i], cnt);
      if (rank) atomicAdd(&s.rank[r.i], rank);
      atomicMin(&s.first[r.i], first);
    }
  }
  __syncthreads();
  {                                                                // where my node's run starts: the elements of the nodes that began earlier
    const int32_t myfirst = r.valid ? s.first[r.i] : 0;
    int slot = 0;
    for (int jb = r.j0; jb < r.j1; jb += 64) {
      const int32_t vj = jb + lane < r.j1 ? s.first[jb + lane] : kNoFirst;
  #pragma unroll
      for (int u = 0; u < 64; ++u) slot += __builtin_amdgcn_readlane(vj, u) < myfirst ? 1 : 0;
    }
    if (mine >= 0 && slot) atomicAdd(&s.slot[r.i], slot);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.B; i += blockDim.x)
    if (s.node[i] >= 0) s.stage[s.slot[i] + s.rank[i]] = s.change[i];
  __syncthreads();
  for (int i = threadIdx.x; i < a.B; i += blockDim.x) {
    if (s.node[i] < 0 || s.rank[i] != 0) continue;                 // the run's first element adds it up
    const double* run = s.stage + s.slot[i];
    const int len = s.cnt[i];
    double acc = treeval;
    int t = 0;
    for (; t + 8 <= len; t += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = run[t + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; t < len; ++t) acc += run[t];
    a.tree[s.node[i]] = acc;
  }
}

__device__ __forceinline__ double block_max(double v, double* sm16) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm16[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sm16[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmax(r, sm16[w]);
  return r;
}

// The maximum over the leaves AS THEY WILL BE after this update: what the NEXT store_transition() gives its new rows
// (:186-189).  Block b scans leaves [b * kMaxChunk, ...): the batch's leaves are skipped (bitmap) and enter with their final
// values (the last element on each leaf), so the scan needs no leaf that this launch writes.  max is exact in any order.
__device__ __forceinline__ void small_max_block(const SmallArgs& a, SmallLds& s, int b, double* sm16) {
  __shared__ uint32_t bm[kMaxChunk / 32];
  const int64_t lo = (int64_t)b * kMaxChunk;
  for (int t = threadIdx.x; t < kMaxChunk / 32; t += blockDim.x) bm[t] = 0u;
  __syncthreads();
  const int i = threadIdx.x;
  int64_t off = -1;
  if (i < a.B) {
    off = (int64_t)s.leaf[i] - (a.cap - 1) - lo;
    if (off >= 0 && off < kMaxChunk) atomicOr(&bm[off >> 5], 1u << (off & 31));
  }
  __syncthreads();
  double v = -1.0e308;
#pragma unroll
  for (int u = 0; u < kMaxChunk / 1024; ++u) {
    const int o = threadIdx.x + 1024 * u;
    if (lo + o < a.cap && !((bm[o >> 5] >> (o & 31)) & 1u)) v = fmax(v, a.tree[a.cap - 1 + lo + o]);
  }
  if (off >= 0 && off < kMaxChunk && s.last[i]) v = fmax(v, s.p[i]);   // (about two elements per block at B = 256, cap = 2^20)
  v = block_max(v, sm16);
  if (threadIdx.x == 0) __hip_atomic_store(a.mx.partial + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(1024) void per_small_kernel(SmallArgs a) {
  __shared__ SmallLds s;
  __shared__ double sm16[kSmallWaves];
  __shared__ unsigned int s_is_last;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const bool depth_block = a.mode == SMALL_ANC || (a.mode == SMALL_FUSED && (int)blockIdx.x < a.depth);
  const bool need_change = a.mode != SMALL_ANC && (a.mode == SMALL_LEAF || depth_block);
  small_load(a, s, need_change);
  if (a.mode == SMALL_ANC)
    for (int i = threadIdx.x; i < a.B; i += blockDim.x) s.change[i] = a.change_ws[i];
  __syncthreads();
  if (a.mode != SMALL_ANC) small_dups(a, s, wave, lane, need_change);   // (a maximum block: the last-writer flags alone)
  if (depth_block) small_ancestors(a, s, (int)blockIdx.x, wave, lane);
  else if (a.mode == SMALL_FUSED) small_max_block(a, s, (int)blockIdx.x - a.depth, sm16);
  if (a.mode == SMALL_ANC) return;
  bool write_leaves = a.mode == SMALL_LEAF;
  if (a.mode == SMALL_LEAF)
    for (int i = threadIdx.x; i < a.B; i += blockDim.x) a.change_ws[i] = s.change[i];
  if (a.mode == SMALL_FUSED) {                                     // the last block to get here: every old leaf value has been read
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      s_is_last = atomicAdd(a.mx.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    write_leaves = s_is_last != 0u;
    if (write_leaves) __threadfence();
  }
  if (!write_leaves) return;
  for (int i = threadIdx.x; i < a.B; i += blockDim.x)
    if (s.last[i]) a.tree[s.leaf[i]] = s.p[i];                     // last writer wins
  if (a.mode == SMALL_FUSED) {
    if (a.mx.nb > 0) {
      double v = -1.0e308;
      for (int t = threadIdx.x; t < a.mx.nb; t += blockDim.x) v = fmax(v, __hip_atomic_load(a.mx.partial + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      v = block_max(v, sm16);
      if (threadIdx.x == 0) a.mx.out[0] = v;
    }
    if (threadIdx.x == 0) *a.mx.ticket = 0u;
  }
}

// ---- the N-row vector store (idx == NULL): per_store_device.hpp ------------------------------------------------------
// The batch tree (pairwise sums over the batch index, P = the next power of two >= B leaves) is built ONCE, by the launch
// that writes the leaves, into the workspace; the ancestor launch is then ~B small independent node updates.  Rounds 3-5 had
// every depth's workgroup rebuild the whole tree in 64-131 KB of LDS: 13 us alone, but 25-30 us in a Rainbow vector step,
// where it runs on the tree's stream beside the acting launch (whose 256 x 16 waves hold every compute unit) and was the end
// of the step's critical chain (profiles/r05_rainbow_timeline_before.txt; after: r05_rainbow_timeline.txt).
constexpr int kSegChunk = 512;                           // batch leaves per workgroup of launch 1
constexpr int kSegTop = per::kStoreChunk / kSegChunk;    // <= 16 chunk roots: the levels above them are rebuilt where needed

// Launch 1: every row's leaf (no duplicates: B <= cap), change = p - old leaf, leaf := p; the chunk's sub-tree of the batch
// tree -> seg_g (node k of the P-leaf tree at seg_g[k], chunk c's root at P / W + c; the leaves stay in change_out).
__global__ __launch_bounds__(kBlock) void per_store_leaf_kernel(double* __restrict__ tree, int64_t cap, int64_t idx_start,
                                                                const int64_t* __restrict__ idx_start_dev, int64_t offset,
                                                                const double* __restrict__ prio,
                                                                const double* __restrict__ ps_dev, double ps, int B, int P,
                                                                double* __restrict__ change_out, double* __restrict__ seg_g) {
  __shared__ double s[2 * kSegChunk];
  if (idx_start_dev) idx_start = idx_start_dev[0];       // recorded into a hipGraph: this replay's ring cursor
  const int W = P < kSegChunk ? P : kSegChunk, c = blockIdx.x;
  for (int i = threadIdx.x; i < W; i += kBlock) {
    const int b = c * W + i;
    double ch = 0.0;                                     // batch positions B .. P - 1 are zeros
    if (b < B) {
      const int64_t leaf = (idx_start + offset + b) % cap + cap - 1;
      const double p = prio_of(prio, ps_dev, ps, b);
      ch = p - tree[leaf];
      tree[leaf] = p;
      change_out[b] = ch;
    }
    s[W + i] = ch;
  }
  __syncthreads();
  for (int w = W >> 1; w >= 1; w >>= 1) {
    for (int k = w + threadIdx.x; k < 2 * w; k += kBlock) s[k] = s[2 * k] + s[2 * k + 1];
    __syncthreads();
  }
  const size_t root = (size_t)(P / W + c);
  for (int k = 1 + threadIdx.x; k < W; k += kBlock) {
    const int w = 1 << (31 - __clz(k));                  // k = w + j: level of width w below the chunk's root
    seg_g[root * w + (k - w)] = s[k];
  }
}

// The batch tree as launch 2 reads it: the levels above the chunk roots from LDS, the rest from seg_g, the leaves in place.
struct StoreSeg {
  const double* top;      // nodes 1 .. C - 1 (C = chunk roots)
  const double* seg;      // nodes C .. P - 1
  const double* leaf;     // change[0 .. B)
  int C, P, B;
  __device__ __forceinline__ double operator[](int k) const {
    return k < C ? top[k] : (k < P ? seg[k] : (k - P < B ? leaf[k - P] : 0.0));
  }
};

// Launch 2: blockIdx.y = node depth d, the depth's nodes over blockIdx.x: <= 4 runs per node in closed form, <= 2 log2(P)
// reads per run, ONE addition into the node.  No chain is longer than log2(P) + 4 float64 adds.
__global__ __launch_bounds__(kBlock) void per_store_ancestor_kernel(double* __restrict__ tree, int64_t cap, int64_t idx_start,
                                                                    const int64_t* __restrict__ idx_start_dev, int64_t offset,
                                                                    const double* __restrict__ change,
                                                                    const double* __restrict__ seg_g, int B, int P) {
  __shared__ double s_top[2 * kSegTop];
  if (idx_start_dev) idx_start = idx_start_dev[0];
  const int d = blockIdx.y;
  const per::StoreGeom g = per::store_geom(cap, idx_start + offset, B);
  int64_t n0[4], n1[4];
  const int nr = per::store_node_ranges(g, d, n0, n1);
  const int64_t first = (int64_t)blockIdx.x * kBlock;
  bool any = false;
  for (int r = 0; r < nr; ++r) any = any || first <= n1[r] - n0[r];
  if (!any) return;                                       // (uniform: before the barrier)
  const int C = P > kSegChunk ? P / kSegChunk : 1;
  if (C > 1) {
    if (threadIdx.x < C) s_top[C + threadIdx.x] = seg_g[C + threadIdx.x];
    __syncthreads();
    for (int w = C >> 1; w >= 1; w >>= 1) {
      if (threadIdx.x < w) s_top[w + threadIdx.x] = s_top[2 * (w + threadIdx.x)] + s_top[2 * (w + threadIdx.x) + 1];
      __syncthreads();
    }
  }
  const StoreSeg seg{s_top, seg_g, change, C, P, B};
  for (int r = 0; r < nr; ++r) {
    const int64_t cnt = n1[r] - n0[r] + 1;
    for (int64_t t = first + threadIdx.x; t < cnt; t += (int64_t)gridDim.x * kBlock) {
      const int64_t node = n0[r] + t;
      bool seen = false;                                   // a node two ranges share belongs to the first
      for (int q = 0; q < r; ++q) seen = seen || (node >= n0[q] && node <= n1[q]);
      if (seen) continue;
      int a[4], e[4];
      const int m = per::store_node_runs(g, d, node, a, e);
      double S = 0.0;
      for (int q = 0; q < m; ++q) S += per::store_run_sum(seg, P, a[q], e[q]);
      tree[node] = tree[node] + S;
    }
  }
}

// ---- large unordered batches (512 < B <= kLdsB with explicit indices) -------------------------------
// The searches above are O(B^2 / threads): 6.7 ms per call at B = 8192.  Ordering the batch by (leaf or ancestor id, batch
// index) makes every node's elements one contiguous run that is already in batch order, so "latest earlier element on the
// same leaf", "last writer" and the ordered per-node sums become neighbour tests and run walks.  The order comes from a
// STABLE block radix sort by the id alone (rocPRIM's block_radix_sort: the batch index is the value, the elements enter in
// index order, so equal ids keep it), over exactly the id's bits — 1 pass for the root's block, 5 for the deepest level of a
// 2^20-leaf tree; rounds 3-4 bitonic-sorted 64-bit (id, index) keys, 91 compare-exchange stages per depth: 593 us per update
// of 8192 indices, 77 % of a Rainbow vector step at that batch.  Keys are rebuilt as id << 13 | index afterwards.
constexpr int kIdxBits = 13;                       // kLdsB = 2^13
constexpr uint64_t kPadKey = ~0ull;
constexpr int kSortItems = kLdsB / 1024;
using BlockSort = rocprim::block_radix_sort<uint32_t, 1024, kSortItems, uint32_t>;

// s_keys[0 .. kLdsB) <- the batch ordered by (id, index); id_of(i) for i < B, pad (= all ones in `bits` bits: above every valid
// id) for "no id"; the pads end up behind the valid elements as kPadKey.  1024 threads; `st` may alias memory the caller uses later.
template <class F>
__device__ __forceinline__ void sort_by_id(uint64_t* s_keys, BlockSort::storage_type& st, int B, unsigned bits, F id_of) {
  const uint32_t pad = bits >= 32 ? 0xFFFFFFFFu : (1u << bits) - 1u;
  uint32_t k[kSortItems], v[kSortItems];
#pragma unroll
  for (int u = 0; u < kSortItems; ++u) {
    const int i = threadIdx.x * kSortItems + u;
    v[u] = (uint32_t)i;
    k[u] = i < B ? id_of(i, pad) : pad;
  }
  BlockSort().sort(k, v, st, 0, bits);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kSortItems; ++u)
    s_keys[threadIdx.x * kSortItems + u] = k[u] == pad ? kPadKey : (((uint64_t)k[u] << kIdxBits) | (uint64_t)v[u]);
  __syncthreads();
}

__global__ __launch_bounds__(1024) void per_leaf_sorted_kernel(double* __restrict__ tree, int64_t cap,
                                                               const int32_t* __restrict__ idx, int idx_is_tree,
                                                               const double* __restrict__ prio,
                                                               const double* __restrict__ ps_dev, double ps, int B, int bits,
                                                               int64_t* __restrict__ leaf_out,
                                                               double* __restrict__ change_out) {
  extern __shared__ uint64_t s_keys[];               // [kLdsB]
  __shared__ BlockSort::storage_type st;
  sort_by_id(s_keys, st, B, (unsigned)bits, [&](int i, uint32_t) {
    const int64_t leaf = idx_is_tree ? (int64_t)idx[i] : (int64_t)idx[i] + cap - 1;
    leaf_out[i] = leaf;
    return (uint32_t)leaf;
  });
  for (int sidx = threadIdx.x; sidx < B; sidx += blockDim.x) {
    const uint64_t key = s_keys[sidx];
    const int64_t leaf = (int64_t)(key >> kIdxBits);
    const int i = (int)(key & ((1u << kIdxBits) - 1));
    // the latest earlier element on the same leaf is the sorted predecessor (same leaf, next smaller index)
    double prev;
    if (sidx > 0 && (int64_t)(s_keys[sidx - 1] >> kIdxBits) == leaf)
      prev = prio_of(prio, ps_dev, ps, (int)(s_keys[sidx - 1] & ((1u << kIdxBits) - 1)));
    else prev = tree[leaf];
    change_out[i] = prio_of(prio, ps_dev, ps, i) - prev;
  }
  __syncthreads();                                   // every old leaf value has been read
  for (int sidx = threadIdx.x; sidx < B; sidx += blockDim.x) {
    const uint64_t key = s_keys[sidx];
    const int64_t leaf = (int64_t)(key >> kIdxBits);
    const bool last = sidx + 1 >= B || (int64_t)(s_keys[sidx + 1] >> kIdxBits) != leaf;
    if (last) tree[leaf] = prio_of(prio, ps_dev, ps, (int)(key & ((1u << kIdxBits) - 1)));   // last writer wins
  }
}

__global__ __launch_bounds__(1024) void per_ancestor_sorted_kernel(double* __restrict__ tree,
                                                                   const int64_t* __restrict__ leaf_g,
                                                                   const double* __restrict__ change_g, int B) {
  extern __shared__ uint64_t s_keys[];               // [kLdsB] keys, then [kLdsB] doubles: the changes IN SORTED ORDER (the sort's scratch first)
  double* s_sorted = reinterpret_cast<double*>(s_keys + kLdsB);
  const int d = blockIdx.x;
  sort_by_id(s_keys, *reinterpret_cast<BlockSort::storage_type*>(s_sorted), B, (unsigned)(d + 1), [&](int i, uint32_t pad) {
    const int64_t lf = leaf_g[i];
    const int L = depth_of(lf);
    return L > d ? (uint32_t)(((lf + 1) >> (L - d)) - 1) : pad;       // (a depth-d node index is < 2^(d+1) - 1 = pad)
  });
  const uint64_t imask = (1u << kIdxBits) - 1;
  // a run's changes side by side, in batch order: its leader then adds contiguous doubles (the root's run is the whole batch —
  // 8192 dependent adds; with a key read, a mask and an indexed read per element that chain was 0.5 ms)
  for (int sidx = threadIdx.x; sidx < B; sidx += blockDim.x) {
    const uint64_t key = s_keys[sidx];
    s_sorted[sidx] = key == kPadKey ? 0.0 : change_g[key & imask];
  }
  __syncthreads();
  for (int sidx = threadIdx.x; sidx < B; sidx += blockDim.x) {
    const uint64_t key = s_keys[sidx];
    if (key == kPadKey) continue;
    const uint64_t node = key >> kIdxBits;
    if (sidx > 0 && (s_keys[sidx - 1] >> kIdxBits) == node) continue;      // not the run's first element
    int lo = sidx, hi = B;                             // the run's end: the first position whose node differs (binary search)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((s_keys[mid] >> kIdxBits) == node) lo = mid; else hi = mid;
    }
    double acc = tree[node];
    int j = sidx;
    for (; j + 8 <= hi; j += 8) {                      // one ordered chain of adds
      double c8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c8[u] = s_sorted[j + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += c8[u];
    }
    for (; j < hi; ++j) acc += s_sorted[j];
    tree[node] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void max_leaf_partial_kernel(const double* __restrict__ tree,
                                                                  int64_t cap,
                                                                  double* __restrict__ partial) {
  double m = -1.0e308;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < cap; i += (int64_t)gridDim.x * kBlock)
    m = fmax(m, tree[cap - 1 + i]);
  __shared__ double sm[kBlock];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(kBlock) void max_final_kernel(const double* __restrict__ partial, int n,
                                                           double* __restrict__ out) {
  double m = -1.0e308;
  for (int i = threadIdx.x; i < n; i += kBlock) m = fmax(m, partial[i]);
  __shared__ double sm[kBlock];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0];
}

__global__ __launch_bounds__(kBlock) void per_priorities_kernel(const float* __restrict__ td, int B,
                                                                double alpha, double eps, double clip,
                                                                double* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  out[i] = td_priority(td[i], alpha, eps, clip);
}

struct PerSampleDev { uint64_t counter; int64_t size; double beta; };
// stratified descent + un-normalised IS weights; per-block max of the weights.
__global__ __launch_bounds__(kBlock) void per_sample_kernel(
    const double* __restrict__ tree, int64_t cap, const double* __restrict__ u, uint64_t seed,
    uint64_t counter, int B, int64_t size, double beta, int variant_b, int32_t* __restrict__ idx_out,
    double* __restrict__ prio_out, float* __restrict__ w32, double* __restrict__ w64,
    double* __restrict__ blockmax, const PerSampleDev* __restrict__ dev, int normalize_here) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int64_t tcap = 2 * cap - 1;
  if (dev) { counter = dev->counter; size = dev->size; beta = dev->beta; }     // this replay's draw scalars
  double w = 0.0;
  if (i < B) {
    const double total = tree[0];
    const double segment = total / (double)B;
    const double a = segment * (double)i, b = segment * (double)(i + 1);
    double ui;
    if (u) ui = u[i];
    else {
      const u32x4 r = philox4x32(seed, (uint32_t)i, 1u, (uint32_t)counter,
                                 RNG_REPLAY | (uint32_t)((counter >> 32) & 0x0FFFFFFFu));
      ui = u01d(r.x, r.y);
    }
    double v = a + (b - a) * ui;                          // np.random.uniform(a, b)
    int64_t p = 0;
    while (true) {
      const int64_t left = 2 * p + 1;
      if (left >= tcap) break;
      const double lv = tree[left];
      if (v <= lv) p = left;
      else { v -= lv; p = left + 1; }
    }
    const double pr = tree[p];
    idx_out[i] = (int32_t)(variant_b ? p : p - cap + 1);
    if (prio_out) prio_out[i] = pr;
    const double prob = pr / total;
    const double wd = pow((double)size * prob, -beta);
    if (variant_b) { w64[i] = wd; w = wd; }
    else { const float wf = (float)wd; w32[i] = wf; w = (double)wf; }   // is_weight is a float32 tensor (:224-226)
  }
  __shared__ double sm[kBlock];
  sm[threadIdx.x] = w;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) blockmax[blockIdx.x] = sm[0];
  if (normalize_here) {        // a single-block draw (B <= 256): per_normalize_kernel's division right here, one launch less
    const double m = fmax(0.0, sm[0]);
    if (i < B) {
      if (variant_b) w32[i] = (float)(w64[i] / m);
      else w32[i] = w32[i] / (float)m;
    }
  }
}

__global__ __launch_bounds__(kBlock) void per_normalize_kernel(float* __restrict__ w32,
                                                               const double* __restrict__ w64, int B,
                                                               const double* __restrict__ blockmax,
                                                               int nblocks, int variant_b) {
  double m = 0.0;
  for (int k = 0; k < nblocks; ++k) m = fmax(m, blockmax[k]);
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  if (variant_b) w32[i] = (float)(w64[i] / m);            // float64 division, then the float32 tensor cast
  else w32[i] = w32[i] / (float)m;                        // is_weight /= is_weight.max() in float32 (:241)
}

}  // namespace

extern "C" {

size_t gymrl_per_workspace_bytes(int B) {
  if (B < 0) return 0;
  const size_t nb = (size_t)cdiv(B > 0 ? B : 1, kBlock);
  return (sizeof(int64_t) + sizeof(double)) * (size_t)B + sizeof(double) * ((size_t)B + nb + 4096) + 256;
}

int gymrl_per_update(double* tree, int64_t cap, const int32_t* idx, int64_t idx_start,
                     int idx_is_tree, const double* prio, const double* prio_scalar_dev,
                     double prio_scalar, int B, const int64_t* idx_start_dev, void* workspace, void* stream_) {
  if (!tree || !workspace || cap <= 0 || B < 0 || (!idx && (idx_start < 0 || B > cap)) || (idx && idx_start_dev)) return -22;
  if (B == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  Ws ws(workspace, B);
  const int use_lds = B <= kLdsB ? 1 : 0;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)per_leaf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsB * 8) != hipSuccess ||
        hipFuncSetAttribute((const void*)per_ancestor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsB * 16) != hipSuccess ||
        hipFuncSetAttribute((const void*)per_leaf_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsB * 8) != hipSuccess ||
        hipFuncSetAttribute((const void*)per_ancestor_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsB * 16) != hipSuccess)
      return -1000 - (int)hipGetLastError();
    attr_set = true;
  }
  // deepest leaf depth = depth of the last tree slot; ancestors live at depths 0 .. that-1
  int depth = 0;
  { int64_t t = 2 * cap - 2; while (t > 0) { t = (t - 1) / 2; ++depth; } }
  if (!idx) {                                           // the N-row vector store: sub-stores of <= 8192 rows, in order
    for (int o = 0; o < B; o += per::kStoreChunk) {
      const int n = B - o < per::kStoreChunk ? B - o : per::kStoreChunk;
      int P = 1;
      while (P < n) P <<= 1;
      // seg_g: P doubles of the scratch area (P <= n + 4096 for every n <= kStoreChunk: gymrl_per_workspace_bytes)
      hipLaunchKernelGGL(per_store_leaf_kernel, dim3(P > kSegChunk ? P / kSegChunk : 1), dim3(kBlock), 0, stream, tree, cap,
                         idx_start, idx_start_dev, (int64_t)o, prio ? prio + o : nullptr, prio_scalar_dev, prio_scalar, n, P,
                         ws.change, ws.partial);
      if (depth > 0)
        hipLaunchKernelGGL(per_store_ancestor_kernel, dim3(cdiv(n / 2 + 2, kBlock), depth), dim3(kBlock), 0, stream, tree, cap,
                           idx_start, idx_start_dev, (int64_t)o, ws.change, ws.partial, n, P);
    }
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  if (B > 512 && B <= kLdsB && 2 * cap < (1ll << 32)) {     // large unordered batch: radix-ordered passes (32-bit ids)
    int bits = 1;
    while ((1ll << bits) - 1 <= 2 * cap - 2) ++bits;        // the largest leaf index 2 cap - 2 stays below the pad (all ones)
    hipLaunchKernelGGL(per_leaf_sorted_kernel, dim3(1), dim3(1024), (size_t)kLdsB * 8, stream, tree, cap, idx, idx_is_tree,
                       prio, prio_scalar_dev, prio_scalar, B, bits, ws.leaf, ws.change);
    if (depth > 0)
      hipLaunchKernelGGL(per_ancestor_sorted_kernel, dim3(depth), dim3(1024), (size_t)kLdsB * 16, stream, tree, ws.leaf,
                         ws.change, B);
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  if (B <= kSmallB && 2 * cap < (1ll << 31)) {                           // the reference's batch sizes: ballots, no searches (two launches: no ticket here)
    SmallArgs a{tree, cap, idx, idx_is_tree, prio, prio_scalar_dev, prio_scalar, B, TdPrio{nullptr, 0.0, 0.0, 0.0}, depth,
                MaxLeaf{0, nullptr, nullptr, nullptr, 0}, ws.change, SMALL_LEAF};
    hipLaunchKernelGGL(per_small_kernel, dim3(1), dim3(1024), 0, stream, a);
    a.mode = SMALL_ANC;
    if (depth > 0) hipLaunchKernelGGL(per_small_kernel, dim3(depth), dim3(1024), 0, stream, a);
    GYMRL_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(per_leaf_kernel, dim3(1), dim3(1024), use_lds ? (size_t)B * 8 : 0, stream, tree, cap, idx,
                     idx_is_tree, prio, prio_scalar_dev, prio_scalar, B, ws.leaf, ws.change, use_lds);
  if (depth > 0)
    hipLaunchKernelGGL(per_ancestor_kernel, dim3(depth), dim3(1024), use_lds ? (size_t)B * 16 : 0, stream, tree,
                       ws.leaf, ws.change, B, use_lds);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_per_max_leaf(const double* tree, int64_t cap, double* out, void* workspace, void* stream_) {
  if (!tree || !out || !workspace || cap <= 0) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  int nb = cdiv(cap, (int64_t)kBlock * 8);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(max_leaf_partial_kernel, dim3(nb), dim3(kBlock), 0, stream, tree, cap, (double*)workspace);
  hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(kBlock), 0, stream, (const double*)workspace, nb, out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_per_priorities(const float* td, int B, double alpha, double eps, double clip, double* prio_out,
                         void* stream_) {
  if (!td || !prio_out || B < 0) return -22;
  if (B == 0) return 0;
  hipLaunchKernelGGL(per_priorities_kernel, dim3(cdiv(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream_,
                     td, B, alpha, eps, clip, prio_out);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

int gymrl_per_update_td(double* tree, int64_t cap, const int32_t* idx, const float* td, int B, double alpha, double eps,
                        double clip, double* max_out, unsigned int* ticket, void* workspace, void* stream_) {
  if (!tree || !idx || !td || !workspace || cap <= 0 || B < 1 || B > kSmallB || 2 * cap >= (1ll << 31) || (max_out && !ticket))
    return -22;
  hipStream_t stream = (hipStream_t)stream_;
  Ws ws(workspace, B);
  int depth = 0;
  { int64_t t = 2 * cap - 2; while (t > 0) { t = (t - 1) / 2; ++depth; } }
  SmallArgs a{tree, cap, idx, 0, nullptr, nullptr, 0.0, B, TdPrio{td, alpha, eps, clip}, depth,
              MaxLeaf{cap, ws.partial, max_out, ticket, 0}, ws.change, SMALL_FUSED};
  const bool fused_max = max_out && cap <= (int64_t)kMaxChunk * 1024;
  if (ticket && depth > 0) {                             // ONE launch: depth blocks | maximum blocks, the last one writes the leaves
    a.mx.nb = fused_max ? cdiv(cap, kMaxChunk) : 0;
    hipLaunchKernelGGL(per_small_kernel, dim3(depth + a.mx.nb), dim3(1024), 0, stream, a);
  } else {
    a.mode = SMALL_LEAF;
    hipLaunchKernelGGL(per_small_kernel, dim3(1), dim3(1024), 0, stream, a);
    a.mode = SMALL_ANC;
    if (depth > 0) hipLaunchKernelGGL(per_small_kernel, dim3(depth), dim3(1024), 0, stream, a);
  }
  GYMRL_CHECK_LAUNCH();
  if (max_out && !(ticket && depth > 0 && fused_max)) return gymrl_per_max_leaf(tree, cap, max_out, ws.partial, stream_);
  return 0;
}

int gymrl_per_sample(const double* tree, int64_t cap, const double* u, uint64_t seed, uint64_t counter,
                     int B, int64_t size, double beta, int variant_b, int32_t* idx_out, double* prio_out,
                     float* w_out, const void* dev, void* workspace, void* stream_) {
  if (!tree || !idx_out || !w_out || !workspace || cap <= 0 || B <= 0 || size <= 0) return -22;
  hipStream_t stream = (hipStream_t)stream_;
  Ws ws(workspace, B);
  const int nb = cdiv(B, kBlock);
  double* w64 = ws.change;                 // reuse: f64[B]
  double* bmax = ws.partial;               // f64[nb]
  hipLaunchKernelGGL(per_sample_kernel, dim3(nb), dim3(kBlock), 0, stream, tree, cap, u, seed, counter, B,
                     size, beta, variant_b, idx_out, prio_out, w_out, w64, bmax, static_cast<const PerSampleDev*>(dev), nb == 1 ? 1 : 0);
  if (nb > 1)
    hipLaunchKernelGGL(per_normalize_kernel, dim3(nb), dim3(kBlock), 0, stream, w_out, w64, B, bmax, nb,
                       variant_b);
  GYMRL_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
