// Standalone probe: shader clock vs 100 MHz wall clock, MFMA f32 16x16x4 issue rate, at
// different grid sizes (how many CUs are busy).  hipcc --offload-arch=gfx950 -O3 clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(long long* out, int iters, float seed) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x + 0] = c1 - c0;
    out[3 * blockIdx.x + 1] = w1 - w0;
    out[3 * blockIdx.x + 2] = (long long)s;
  }
}

__global__ void probe_valu(long long* out, int iters, float seed) {
  float x = seed + threadIdx.x, y = seed;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, y, 1.0f);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x + 0] = c1 - c0;
    out[3 * blockIdx.x + 1] = w1 - w0;
    out[3 * blockIdx.x + 2] = (long long)x;
  }
}

typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void probe_pk(long long* out, int iters, float seed) {
  v2f x = {seed + threadIdx.x, seed * 0.25f}, y = {seed, seed * 0.5f}, one = {1.0f, 1.0f};
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) { x = x * y; x = x + one; }      // v_pk_mul_f32 -> v_pk_add_f32, dependent
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)(x[0] + x[1]); }
}
__global__ void probe_muladd(long long* out, int iters, float seed) {
  float x = seed + threadIdx.x, y = seed;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) { x = x * y; x = x + 1.0f; }      // v_mul_f32 -> v_add_f32, dependent (contract off)
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
__global__ void probe_muladd2(long long* out, int iters, float seed) {
  float x = seed + threadIdx.x, y = seed, u = seed * 0.5f + threadIdx.x;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) { x = x * y; u = u * y; x = x + 1.0f; u = u + 1.0f; }   // two independent chains
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)(x + u); }
}

int main() {
  long long* d;
  hipMalloc(&d, sizeof(long long) * 3 * 4096);
  std::vector<long long> h(3 * 4096);
  for (int waves_per_wg : {1, 4}) {
    for (int grid : {1, 64, 256, 1024}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64 * waves_per_wg), 0, 0, d, 256, 1.0f);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), d, sizeof(long long) * 3 * grid, hipMemcpyDeviceToHost);
      double clk = (double)h[0], wall = (double)h[1];
      // 256 iters * 64 mfma
      printf("mfma  grid %4d x %d waves: shader clocks %.0f, wall(100MHz) %.0f -> %.3f GHz, %.1f clk/mfma, %.2f us\n", grid,
             waves_per_wg, clk, wall, clk / wall * 0.1, clk / (256.0 * 64), wall / 100.0);
    }
  }
  for (int grid : {1, 256, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(probe_valu, dim3(grid), dim3(64), 0, 0, d, 256, 1.0f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, sizeof(long long) * 3 * grid, hipMemcpyDeviceToHost);
    double clk = (double)h[0], wall = (double)h[1];
    printf("valu  grid %4d x 1 wave: shader clocks %.0f, wall %.0f -> %.3f GHz, %.2f clk/fma(dependent), %.2f us\n", grid, clk,
           wall, clk / wall * 0.1, clk / (256.0 * 64), wall / 100.0);
  }
  {
    hipLaunchKernelGGL(probe_pk, dim3(1), dim3(64), 0, 0, d, 256, 1.0f); hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(long long) * 3, hipMemcpyDeviceToHost);
    printf("pk_mul->pk_add dependent: %.2f clk per instruction\n", (double)h[0] / (256.0 * 64));
    hipLaunchKernelGGL(probe_muladd, dim3(1), dim3(64), 0, 0, d, 256, 1.0f); hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(long long) * 3, hipMemcpyDeviceToHost);
    printf("mul->add dependent (scalar): %.2f clk per instruction\n", (double)h[0] / (256.0 * 64));
    hipLaunchKernelGGL(probe_muladd2, dim3(1), dim3(64), 0, 0, d, 256, 1.0f); hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(long long) * 3, hipMemcpyDeviceToHost);
    printf("two independent mul->add chains: %.2f clk per instruction\n", (double)h[0] / (256.0 * 128));
  }
  return 0;
}
