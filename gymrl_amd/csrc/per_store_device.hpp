// per_store_device.hpp — the N-row VECTOR STORE of the prioritised-replay sum tree in closed form.
//
// PrioritizedNStepBuffer.store_transition (rainbow_dqn_cartpole.py:179-205) stores ONE row per env step; a vector step
// stores N rows at consecutive ring positions with one priority (1.0 or priority_max).  The order in which those N
// changes reach an ancestor is therefore this build's own definition (include/gymrl.h, "vector store"): a node's elements
// — the batch indices b whose leaf lies below it — are grouped into maximal runs of consecutive b, a run is summed from the
// canonical blocks of a complete binary tree over the batch index, the runs are added in ascending order and the node
// receives ONE addition.  At N = 1 that is the reference's `tree[parent] += change` (:122-128) bit for bit; at N = 8192 the
// root no longer waits for an 8192-long dependent chain of float64 adds (round 3: 72-79 us per store, 0.004 of HBM).
//
// Everything here is integer geometry shared by the kernels in per.hip and — compiled with g++, no HIP — by the CPU test
// tests/test_per_store_geometry.py, which checks these closed forms against member lists built by walking every leaf's
// ancestors.  Leaves of a tree with a non-power-of-two capacity sit at two depths (the array rule is followed literally),
// a ring range may wrap: a node owns up to four ranges of b, adjacent ones merge into one run.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GYMRL_PER_HD __host__ __device__ __forceinline__
#else
#define GYMRL_PER_HD inline
#endif

namespace gymrl {
namespace per {

constexpr int kStoreChunk = 8192;       // rows per sub-store: the batch tree (2 * 8192 doubles) lives in LDS

struct StoreGeom {
  int64_t cap;      // leaves
  int64_t s;        // ring position of batch element 0, in [0, cap)
  int B;            // rows, 1 <= B <= min(cap, kStoreChunk)
  int Lmax;         // depth of the deepest leaves (root = 0)
  int64_t deep0;    // first tree position at depth Lmax = 2^Lmax - 1; positions [cap - 1, deep0) are the shallower leaves
};

GYMRL_PER_HD StoreGeom store_geom(int64_t cap, int64_t start, int B) {
  StoreGeom g;
  g.cap = cap; g.s = start % cap; g.B = B;
  int L = 0;
  for (int64_t t = 2 * cap - 2; t > 0; t = (t - 1) / 2) ++L;
  g.Lmax = L;
  g.deep0 = ((int64_t)1 << L) - 1;
  return g;
}

// The ancestors at depth d of the batch's leaves: up to four inclusive node ranges (two ring pieces x two leaf depths),
// possibly overlapping.  Returns their number.
GYMRL_PER_HD int store_node_ranges(const StoreGeom& g, int d, int64_t (&n0)[4], int64_t (&n1)[4]) {
  int nr = 0;
  const int64_t end = g.s + g.B;                                    // one past the last ring position, before wrapping
  for (int piece = 0; piece < 2; ++piece) {
    int64_t x0, x1;                                                 // data-index range of this ring piece
    if (piece == 0) { x0 = g.s; x1 = (end < g.cap ? end : g.cap) - 1; }
    else { if (end <= g.cap) break; x0 = 0; x1 = end - g.cap - 1; }
    const int64_t p0 = x0 + g.cap - 1, p1 = x1 + g.cap - 1;         // tree positions
    for (int deep = 0; deep < 2; ++deep) {
      const int L = deep ? g.Lmax : g.Lmax - 1;
      int64_t q0 = p0, q1 = p1;
      if (deep) { if (q0 < g.deep0) q0 = g.deep0; }
      else { if (q1 > g.deep0 - 1) q1 = g.deep0 - 1; }
      if (q0 > q1 || L <= d) continue;
      n0[nr] = ((q0 + 1) >> (L - d)) - 1;
      n1[nr] = ((q1 + 1) >> (L - d)) - 1;
      ++nr;
    }
  }
  return nr;
}

// The runs [a, e) of batch indices below `node` (depth d), ascending and maximal.  Returns their number (>= 1 for a node
// that store_node_ranges produced).
GYMRL_PER_HD int store_node_runs(const StoreGeom& g, int d, int64_t node, int (&a)[4], int (&e)[4]) {
  int n = 0;
  for (int deep = 0; deep < 2; ++deep) {
    const int L = deep ? g.Lmax : g.Lmax - 1;
    const int k = L - d;
    if (k < 1) continue;
    int64_t q0 = ((node + 1) << k) - 1, q1 = ((node + 2) << k) - 2;             // the node's descendants at depth L
    const int64_t lo = deep ? g.deep0 : g.cap - 1, hi = deep ? 2 * g.cap - 2 : g.deep0 - 1;   // the leaves among them
    if (q0 < lo) q0 = lo;
    if (q1 > hi) q1 = hi;
    if (q0 > q1) continue;
    const int64_t x0 = q0 - (g.cap - 1), x1 = q1 - (g.cap - 1);                 // data indices
    // ring positions at or after s: b = x - s
    { const int64_t xa = x0 > g.s ? x0 : g.s;
      if (xa <= x1) { int64_t b0 = xa - g.s, b1 = x1 - g.s; if (b1 > g.B - 1) b1 = g.B - 1; if (b0 <= b1) { a[n] = (int)b0; e[n] = (int)b1 + 1; ++n; } } }
    // ring positions before s (the range wrapped): b = x - s + cap
    { const int64_t xb = x1 < g.s - 1 ? x1 : g.s - 1;
      if (x0 <= xb) { int64_t b0 = x0 - g.s + g.cap, b1 = xb - g.s + g.cap; if (b1 > g.B - 1) b1 = g.B - 1; if (b0 <= b1) { a[n] = (int)b0; e[n] = (int)b1 + 1; ++n; } } }
  }
  for (int i = 1; i < n; ++i)                                       // ascending (insertion sort of <= 4 disjoint ranges)
    for (int j = i; j > 0 && a[j] < a[j - 1]; --j) { int t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; t = e[j]; e[j] = e[j - 1]; e[j - 1] = t; }
  int m = 0;                                                        // adjacent ranges are ONE run
  for (int i = 0; i < n; ++i) {
    if (m > 0 && a[i] == e[m - 1]) e[m - 1] = e[i];
    else { a[m] = a[i]; e[m] = e[i]; ++m; }
  }
  return m;
}

// One run from the canonical blocks of the batch tree seg[1 .. 2P) (leaves at seg[P + b]).
template <class SegPtr>
GYMRL_PER_HD double store_run_sum(SegPtr seg, int P, int a, int e) {
  double sl = 0.0, sr = 0.0;
  for (int l = a + P, r = e + P; l < r; l >>= 1, r >>= 1) {
    if (l & 1) sl += seg[l++];
    if (r & 1) sr += seg[--r];
  }
  return sl + sr;
}

}  // namespace per
}  // namespace gymrl
