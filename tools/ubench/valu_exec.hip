// Micro-benchmark: cadence of a dependent f32 FMA chain of ONE wave as a function of the active lanes (does gfx950 skip
// the 16-lane passes whose EXEC bits are all zero?) and of the chain count (issue- vs latency-bound).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_exec valu_exec.hip ; run: ./valu_exec
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void chain(float* out, long long* cyc, int active, int iters) {
  int lane = threadIdx.x;
  float a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) a[c] = 1.0f + lane * 1e-3f + c;
  float m = 0.999f + out[0], b = 1e-3f;
  long long t0 = 0, t1 = 0;
  if (lane < active) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fmaf(a[c], m, b);
    }
    t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    out[1 + lane] = s;
  }
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int CHAINS>
void run(int active) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 128); hipMalloc(&cyc, 8);
  hipMemset(out, 0, 4 * 128);
  int iters = 20000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  chain<CHAINS><<<1, 64>>>(out, cyc, active, 100);
  hipEventRecord(s);
  chain<CHAINS><<<1, 64>>>(out, cyc, active, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * CHAINS;
  printf("chains %d active %2d: %.2f ns per FMA instruction, %.2f counter ticks per instruction\n", CHAINS, active,
         ms * 1e6 / n, (double)c / n);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int act : {64, 48, 32, 16, 4}) run<1>(act);
  for (int act : {64, 32, 16}) run<2>(act);
  for (int act : {64, 16}) run<4>(act);
  return 0;
}
