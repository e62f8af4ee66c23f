// Standalone probe: does a wave64 VALU instruction get cheaper when only the first 16 / 32 lanes are enabled?
// 8 independent v_mul/v_add chains (issue-bound, not latency-bound), one wave per workgroup.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exec_probe.hip -o exec_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(long long* out, int iters, float seed, int active) {
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = seed + threadIdx.x + k;
  const float y = seed;
  long long c0 = 0, c1 = 0;
  if ((int)threadIdx.x < active) {
    c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = x[k] * y; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = x[k] + 1.0f; }
      }
    }
    c1 = clock64();
  }
  float s = 0;
  for (int k = 0; k < 8; ++k) s += x[k];
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = (long long)s; }
}

int main() {
  long long* d;
  hipMalloc(&d, 64);
  long long h[2];
  const int iters = 2000;
  for (int active : {64, 48, 32, 16, 8, 4, 1}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, iters, 1.0001f, active);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, iters, 1.0001f, active);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("active lanes %2d: %.2f clk per VALU instruction (independent chains)\n", active, (double)h[0] / (iters * 128.0));
  }
  return 0;
}
