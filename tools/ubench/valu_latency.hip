// Micro-benchmark: dependent-issue latency (one wave, one chain) and throughput (independent chains) of the VALU
// instruction kinds the LunarLander solver is made of, on gfx950.  One wave on one SIMD — the situation of the
// persistent rollout kernel's solver wave.  build: hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ void lat(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  float a = 1.0f + lane * 1e-3f, b = 1.5f - lane * 1e-3f, c = 0.7f + lane * 1e-4f, d = 1.1f;
  const float m = 0.999f + out[0], k = 1e-3f + out[0];
  v2f pa = {a, b}, pm = {m, m}, pk = {k, k};
  unsigned long long msk = __builtin_amdgcn_ballot_w64((lane & 1) != 0), msk2 = __builtin_amdgcn_ballot_w64((lane & 2) != 0); float sc = 1.0f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));) }
    if (KIND == 1) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(k));) }
    if (KIND == 2) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));) }
    if (KIND == 3) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa) : "v"(pm));) }
    if (KIND == 4) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa) : "v"(pm), "v"(pk));) }
    if (KIND == 5) { REP16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(k));) }
    if (KIND == 6) { REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(k), "v"(m) : "vcc");) }
    if (KIND == 7) { REP16(a = __builtin_amdgcn_mov_dpp(a, 0x00, 0xf, 0xf, true) * m;) }     // quad_perm [0,0,0,0] + mul
    if (KIND == 8) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a));) }
    if (KIND == 9) { REP16(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));) }
    if (KIND == 10) { REP16(a = a / m;) }                                                    // IEEE division
    if (KIND == 11) { REP16(a = __builtin_sqrtf(a) + k;) }                                    // correctly rounded sqrt + add
    // throughput: 2 / 4 independent chains of mul
    if (KIND == 12) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 13) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (KIND == 14) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(m), "v"(k));) }
    if (KIND == 15) { REP16(asm volatile("v_mov_b32 %0, %0" : "+v"(a));) }
    if (KIND == 16) { REP16(asm volatile("v_mul_f32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(m));) }
    if (KIND == 17) { REP16(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(m), "v"(k));) }
    if (KIND == 18) { REP16(asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(a) : "v"(m));) }   // VOP3 encoding
    if (KIND == 20) { REP16(asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %1 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(m));) }
    if (KIND == 21) { REP16(asm volatile("v_mul_f32_dpp %0, %2, %3 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %2, %3 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b) : "v"(c), "v"(m));) }
    if (KIND == 22) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32_dpp %0, %0, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 23) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 24) { REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %0, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
    if (KIND == 25) { REP16(asm volatile("s_nop 1" ::);) }
    if (KIND == 26) { REP16(asm volatile("s_nop 0" ::);) }
    if (KIND == 27) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "s"(msk));) }
    if (KIND == 29) { REP16(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(k), "v"(m));) }
    if (KIND == 30) { REP16(asm volatile("v_mul_f32_dpp %0, %0, %1 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(m));) }
    if (KIND == 31) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_readlane_b32 %1, %0, 2\n v_mul_f32 %0, %0, %1" : "+v"(a), "+s"(sc) : "v"(m));) }
    if (KIND == 19) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %3" : "+v"(b), "+v"(pa) : "v"(m), "v"(pm));) }
  }
  long long t1 = __builtin_readcyclecounter();
  out[1 + lane] = a + b + c + d + pa.x + pa.y;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int KIND>
void run(const char* name, int per16) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4 * 128); (void)hipMalloc(&cyc, 8);
  (void)hipMemset(out, 0, 4 * 128);
  const int iters = 4000;
  printf("%-52s ", name); fflush(stdout);
  lat<KIND><<<1, 64>>>(out, cyc, 50);
  lat<KIND><<<1, 64>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%6.2f cycles per step (%d instr per step)\n", (double)c / ((double)iters * 16), per16); fflush(stdout);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_mul_f32 chain", 1); run<1>("v_add_f32 chain", 1); run<2>("v_fma_f32 chain", 1); run<17>("v_fmac_f32 chain", 1);
  run<18>("v_mul_f32_e64 (VOP3, neg) chain", 1);
  run<14>("v_mul -> v_add chain", 2);
  run<3>("v_pk_mul_f32 chain", 1); run<4>("v_pk_fma_f32 chain", 1);
  run<5>("v_max_f32 chain", 1); run<6>("v_cmp + v_cndmask chain", 2); run<7>("quad-broadcast DPP mov + mul chain", 2);
  run<15>("v_mov_b32 chain", 1);
  run<8>("v_rcp_f32 chain", 1); run<9>("v_sqrt_f32 chain", 1); run<10>("IEEE a / m chain", 1); run<11>("sqrtf(a) + k chain", 2);
  run<16>("v_mul + s_nop 0 chain", 2);
  run<20>("s_nop 1 + dependent v_mul_f32_dpp chain", 2); run<30>("dependent v_mul_f32_dpp chain, no nop (hw interlock?)", 1);
  run<21>("2 independent v_mul_f32_dpp (src ready)", 2);
  run<22>("mul a, mul b, dpp-mul a (1 instr between)", 3); run<23>("mul a, mul b, mul a (reference)", 3);
  run<24>("s_nop 1 + v_mov_dpp + v_mul chain", 3); run<25>("s_nop 1 alone", 1); run<26>("s_nop 0 alone", 1);
  run<27>("v_cndmask_b32_e64 chain (sgpr mask)", 1); run<29>("v_med3_f32 chain", 1);
  run<31>("mul, v_readlane, mul (sgpr) chain", 3);
  run<12>("2 independent v_mul chains", 2); run<13>("4 independent v_mul chains", 4); run<19>("v_mul + v_pk_mul independent", 2);
  return 0;
}
