// finalize_device.hpp — the second halves of the update path's batch reductions as device functions, shared by the kernels
// that run them alone (gemm.hip: tn_reduce_kernel / tn_colsum_kernel behind gymrl_linear_bwd_weight) and by the ONE launch
// that closes a minibatch's backward pass (mlp_train.hip: gymrl_update_finalize).  Same loads, same float64 sums in the same
// order either way: `block` is the position inside the job's own grid.
#pragma once
#include "train_device.hpp"

namespace gymrl {
namespace fin {

constexpr int kCUs = 256;
constexpr size_t kColsumBytes = (size_t)kCUs * 512 * sizeof(float);    // first part of gymrl_linear_bwd_weight's workspace: column-sum partials

// the weight-gradient kernel's cut of the batch (gymrl_linear_bwd_weight_geometry)
__host__ __device__ __forceinline__ void tn_geometry(int64_t B, int N, int* slices, int64_t* rps) {
  const int ntiles = N / 256;
  int s = kCUs / ntiles;
  int64_t r = (B + s - 1) / s;
  r += r & 1;                                     // row pairs are the MFMA's K = 2
  if (r < 64) r = 64;                             // tiny minibatches: fewer, non-trivial slices
  s = (int)((B + r - 1) / r);
  if (s < 1) s = 1;
  *slices = s;
  *rps = r;
}

// db[ntile*256 + n] from the per-slice column sums, same slice grouping as the weight tiles: thread (n, j) adds the
// slices s = j (mod 4) ascending (independent loads, 8 in flight), the four group sums combine as ((g0 + g1) + g2) + g3.
// 256 threads, ntiles * 4 blocks; sm: double[4][64].
__device__ __forceinline__ void tn_colsum_body(const float* __restrict__ parts, int slices, float* __restrict__ db, int block,
                                               double (*sm)[64]) {
  const int n = block * 64 + (threadIdx.x & 63), j = threadIdx.x >> 6;     // n: column over all tiles
  const int ntile = n >> 8, nl = n & 255;
  const float* src = parts + (size_t)ntile * slices * 256 + nl;
  double g = 0.0;
  int s = j;
  for (; s + 28 < slices; s += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(s + 4 * k) * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) g += (double)v[k];
  }
  for (; s < slices; s += 4) g += (double)src[(size_t)s * 256];
  sm[j][threadIdx.x & 63] = g;
  __syncthreads();
  if (j == 0) {
    const int l = threadIdx.x;
    db[n] = (float)(((sm[0][l] + sm[1][l]) + sm[2][l]) + sm[3][l]);
  }
}

// dW[ntile*256 + n][k] = ((g0 + g1) + g2) + g3, g_j = sum over slices s = j (mod 4) ascending (f64).
// 256 threads, ntiles * 256 blocks; sm: double[3][64][4].
__device__ __forceinline__ void tn_reduce_body(const float* __restrict__ parts, int slices, int ldw, float* __restrict__ dW,
                                               int block, double (*sm)[64][4]) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int g = threadIdx.x >> 6, e = block * 64 + (threadIdx.x & 63);    // e: float4 index inside all tiles
  const int ntile = e >> 14, e4 = e & 16383;
  const f32x4* src = reinterpret_cast<const f32x4*>(parts) + (size_t)ntile * slices * 16384 + e4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int s = g;
  for (; s + 12 < slices; s += 16) {
    const f32x4 v0 = src[(size_t)s * 16384], v1 = src[(size_t)(s + 4) * 16384];
    const f32x4 v2 = src[(size_t)(s + 8) * 16384], v3 = src[(size_t)(s + 12) * 16384];
    s0 += (double)v0[0]; s1 += (double)v0[1]; s2 += (double)v0[2]; s3 += (double)v0[3];
    s0 += (double)v1[0]; s1 += (double)v1[1]; s2 += (double)v1[2]; s3 += (double)v1[3];
    s0 += (double)v2[0]; s1 += (double)v2[1]; s2 += (double)v2[2]; s3 += (double)v2[3];
    s0 += (double)v3[0]; s1 += (double)v3[1]; s2 += (double)v3[2]; s3 += (double)v3[3];
  }
  for (; s < slices; s += 4) {
    const f32x4 v = src[(size_t)s * 16384];
    s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
  }
  const int le = threadIdx.x & 63;
  if (g > 0) { sm[g - 1][le][0] = s0; sm[g - 1][le][1] = s1; sm[g - 1][le][2] = s2; sm[g - 1][le][3] = s3; }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { s0 += sm[j][le][0]; s1 += sm[j][le][1]; s2 += sm[j][le][2]; s3 += sm[j][le][3]; }
    const int n = e4 >> 6, k4 = e4 & 63;
    *reinterpret_cast<f32x4*>(dW + (size_t)(ntile * 256 + n) * ldw + 4 * k4) = f32x4{(float)s0, (float)s1, (float)s2, (float)s3};
  }
}

}  // namespace fin
}  // namespace gymrl
