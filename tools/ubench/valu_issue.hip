// Clean issue / latency numbers for ONE wave on gfx950: every case is a single asm block of 8 instructions (the compiler
// puts an s_nop between separate asm statements, which hid the difference in valu_latency.hip), looped 16 x per iteration.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define R4(X) X X X X
template <int KIND>
__global__ void k(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  float a = 1.0f + lane * 1e-3f, b = 1.5f, c = 0.7f, d = 1.1f, e = 1.2f, f = 1.3f, g = 1.4f, h = 1.6f;
  const float m = 0.999f + out[0];
  v2f pa = {a, b}, pb = {c, d}, pc = {e, f}, pd = {g, h}, pm = {m, m};
  double da = a, db = b, dc = c, dd = d; const double dm = m;
  const unsigned long long msk = __builtin_amdgcn_ballot_w64((lane & 1) != 0);
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { R4(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));) }
    if (KIND == 1) { R4(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(m));) }
    if (KIND == 2) { R4(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 3) { R4(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1" : "+v"(pa) : "v"(pm));) }
    if (KIND == 4) { R4(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pm));) }
    if (KIND == 5) { R4(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "s"(msk));) }
    if (KIND == 6) { R4(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0" ::);) }
    if (KIND == 7) { R4(asm volatile("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1" ::);) }
    if (KIND == 8) { R4(asm volatile("v_mul_f32 %0, %0, %2\n s_nop 1\n v_mul_f32_dpp %0, %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_mul_f32_dpp %0, %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_mul_f32_dpp %0, %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 9) { R4(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32_dpp %0, %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %1, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %0, %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %1, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 10) { R4(asm volatile("v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_mul_f32 %2, %2, %3\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_mul_f32 %2, %2, %3\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(m));) }
    if (KIND == 11) { R4(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(iters) :: "scc"); iters -= 8;) }
    if (KIND == 13) { R4(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1" : "+v"(da) : "v"(dm));) }
    if (KIND == 14) { R4(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) : "v"(dm));) }
    if (KIND == 12) { R4(asm volatile("v_mul_f32 %0, %0, %1\n s_nop 0\n v_mul_f32 %0, %0, %1\n s_nop 0\n v_mul_f32 %0, %0, %1\n s_nop 0\n v_mul_f32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(m));) }
  }
  long long t1 = __builtin_readcyclecounter();
  out[1 + lane] = (float)(da + db + dc + dd) + a + b + c + d + e + f + g + h + pa.x + pa.y + pb.x + pb.y + pc.x + pc.y + pd.x + pd.y;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int KIND>
void run(const char* name) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4 * 128); (void)hipMalloc(&cyc, 8);
  (void)hipMemset(out, 0, 4 * 128);
  const int iters = 4000;
  printf("%-64s ", name); fflush(stdout);
  k<KIND><<<1, 64>>>(out, cyc, 50);
  k<KIND><<<1, 64>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%6.2f cycles per instruction\n", (double)c / ((double)iters * 32)); fflush(stdout);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_mul_f32, each depends on the previous");
  run<2>("v_mul_f32, two interleaved chains");
  run<10>("v_mul_f32, three interleaved chains");
  run<1>("v_mul_f32, eight independent");
  run<3>("v_pk_mul_f32, each depends on the previous");
  run<4>("v_pk_mul_f32, four interleaved chains");
  run<5>("v_cndmask_b32 (SGPR mask), each depends on the previous");
  run<6>("s_nop 0"); run<7>("s_nop 1");
  run<12>("v_mul_f32 dependent + s_nop 0 (per pair / 2)");
  run<8>("dependent v_mul_f32_dpp with s_nop 1 (8 slots: 5 VALU + 3 nops)");
  run<9>("two interleaved chains incl. v_mul_f32_dpp, no nops");
  run<11>("s_add_u32 dependent");
  run<13>("v_add_f64, each depends on the previous"); run<14>("v_add_f64, four interleaved chains");
  return 0;
}
