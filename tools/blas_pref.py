"""(tools only — A/B baselines; the product has no library GEMM on its paths since round 3.)
Library selection for the small-M float32 GEMMs of the round-1 off-policy updates.

The DQN / Rainbow / SAC networks are 256-wide MLPs updated on 128-256 row minibatches.  hipBLASLt's default
heuristic answers those shapes with its 256 x 128 macro tile: a 128 x 256 x 256 GEMM becomes TWO workgroups on
two of the 256 CUs and takes 27-36 us — the same as at 4096 rows (rocprofv3:
profiles/r01_sac_graph_kernel_stats.csv, 59 GEMMs = 0.93 ms of a 1.5 ms SAC step).  rocBLAS answers the same
shapes in 6-15 us (and `nn.SmallLinear` keeps the bias out of the GEMM call, because addmm's fused-bias form goes
to hipBLASLt whatever the preference).  `small_gemm_backend()` scopes that preference (and, opt-in, PyTorch's TunableOp search
over both libraries) to the off-policy train() loops; the PPO path's 262 144-row GEMMs keep the default
(hipBLASLt at 0.85 of the f32 MFMA peak).  A hipGraph captured inside the scope keeps the kernels chosen in it.
"""
import contextlib

import torch


@contextlib.contextmanager
def small_gemm_backend(prefer="rocblas", tune=False):
    """prefer: "rocblas" | "hipblaslt" | "default".  tune: let TunableOp time every candidate per GEMM shape the
    first time it is seen (a few seconds at start-up; the choice then varies run to run, so bit-reproducibility
    across processes is lost — within a process it is kept)."""
    prev = torch.backends.cuda.preferred_blas_library()
    name = {"rocblas": "cublas", "hipblaslt": "cublaslt", "default": "default"}[prefer]
    torch.backends.cuda.preferred_blas_library(name)
    was_on, was_tuning = torch.cuda.tunable.is_enabled(), torch.cuda.tunable.tuning_is_enabled()
    if tune:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(15)
        torch.cuda.tunable.set_max_tuning_iterations(20)
    try:
        yield
    finally:
        torch.backends.cuda.preferred_blas_library(prev)
        if tune:
            torch.cuda.tunable.tuning_enable(was_tuning)
            torch.cuda.tunable.enable(was_on)
