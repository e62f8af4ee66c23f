// What shader clock does a latency-bound persistent kernel run at?  256 workgroups x 256 threads, only wave 0 of each
// works (a dependent VALU chain), like the persistent rollout kernel's solver wave; compares the shader-clock counter
// with the event time.  build: hipcc --offload-arch=gfx950 -O3 -o clock_under_load clock_under_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* out, long long* cyc, long long* wall, int iters, int waves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a = 1.0f + lane * 1e-3f;
  const float m = 0.999f + out[0];
  long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  if (wave < waves) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 64; ++u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    }
  }
  long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (a == 123.0f) out[1] = a;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}
int main() {
  float* out; long long *cyc, *wall;
  (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 8 * 1024); (void)hipMalloc(&wall, 8 * 1024);
  (void)hipMemset(out, 0, 1024);
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  for (int blocks : {1, 256, 512}) for (int waves : {1, 4}) for (int iters : {20000, 400000}) {
    spin<<<blocks, 256>>>(out, cyc, wall, 100, waves);
    (void)hipEventRecord(s);
    spin<<<blocks, 256>>>(out, cyc, wall, iters, waves);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    long long c, w; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
    printf("blocks %3d waves %d: %8.2f ms, %.2f cycles per v_mul, shader clock %.0f MHz (counter / event time), wall counter %.1f MHz\n",
           blocks, waves, ms, (double)c / ((double)iters * 64), c / (ms * 1e3), w / (ms * 1e3));
    fflush(stdout);
  }
  return 0;
}
