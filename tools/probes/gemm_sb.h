/* gemm_sb.h — entry points of the split-bf16 PROBE library (tools/probes/libgymrl_probe_sb.so).  Not part of the product
 * ABI (include/gymrl.h): built and measured in rounds 3-4 (DESIGN.md sections 4a / 5: 245 us against the 200 us kill line), kept
 * here with its tests and micro-benchmarks so that nobody measures it again. */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/*
 * OPT-IN split-bf16 variants of the forward and the input gradient above for the 256-wide layers (K = 256, N in {256, 512}): csrc/gemm_sb.hip.
 * Every f32 operand is split exactly into three bf16 pieces (hi / mid / lo by truncation) and six bf16 MFMAs per 16-deep step
 * (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; f32 accumulation) reproduce the f32 products to ~1.2e-7 relative at 6/16 of
 * the exact f32-MFMA time.  f32-ACCURATE, not bit-exact: results are compared with float64 (error not above the exact
 * kernels' on benign and adversarial inputs, tests/test_gemm_sb_gpu.py), never with the oracle's fmaf chain.  TWO entry points,
 * no weight-gradient variant; no trainer selects them (there is no Config switch and no bench.py line): they are reached only
 * from tests/test_gemm_sb_gpu.py and tools/{micro,abl,pmc}_gemm_sb.py — built and measured, not enabled.
 * Same argument meaning as the exact entry points.
 */
int gymrl_linear_fwd_sb(const float* X, const float* W, const float* b, int64_t B, int K, int N, int act,
                        float* Y, void* stream);
/* Round 4's bounded experiment on that mode: the activations split by their PRODUCER.  gymrl_split_planes writes the three bf16
 * planes of an f32 array (P[plane][n], 6 bytes per element; in a pipeline the producing layer's epilogue would); the consumer
 * gymrl_linear_fwd_sb_planes is gymrl_linear_fwd_sb reading them ([3][B][256] bf16) — MFMAs, LDS reads and loads only, the
 * same products in the same order (bit-identical results).  Measured and not adopted: DESIGN.md section 4a. */
int gymrl_split_planes(const float* X, int64_t n, void* planes, void* stream);
int gymrl_linear_fwd_sb_planes(const void* X_planes, const float* W, const float* b, int64_t B, int K, int N, int act, float* Y,
                               void* stream);
int gymrl_linear_bwd_input_sb(const float* dY, const float* W, const float* H, int64_t B, int N, int K,
                              float* dX, void* stream);
#ifdef __cplusplus
}
#endif
