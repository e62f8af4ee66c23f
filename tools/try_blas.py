#!/usr/bin/env python3
"""A/B in one process: the SAC / Rainbow vector step under different GEMM back ends for the small-M GEMMs
(hipBLASLt default heuristics, rocBLAS, TunableOp-selected solutions)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import rainbow_dqn_cartpole, sac_pendulum  # noqa: E402


def run(kind, steps=300, warm=60):
    if kind == "sac":
        c = sac_pendulum.Config()
        c.num_envs, c.memory_capacity, c.max_episodes, c.batch_size = 4096, 1 << 20, 10**9, 128
        tr = sac_pendulum.SACTrainer(c)
    else:
        c = rainbow_dqn_cartpole.Config()
        c.num_envs, c.memory_capacity, c.max_episodes, c.batch_size = 8192, 1 << 20, 10**9, 256
        tr = rainbow_dqn_cartpole.RainbowDQNTrainer(c)
    out = sys.stdout
    sys.stdout = open(os.devnull, "w")
    tr.train(max_vector_steps=warm)
    torch.cuda.synchronize()
    tr.env = type(tr.env)(tr.cfg.env_name, tr.cfg.num_envs, device=tr.device, seed=tr.base_seed)
    t0 = time.perf_counter()
    tr.train(max_vector_steps=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sys.stdout = out
    return dt / steps * 1e3


mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
elif mode in ("tunable", "tunable_rocblas"):
    if mode == "tunable_rocblas":
        torch.backends.cuda.preferred_blas_library("cublas")
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(15)
    torch.cuda.tunable.set_max_tuning_iterations(20)
    torch.cuda.tunable.set_filename("/tmp/tunable_gymrl.csv")
print(mode, {k: round(run(k), 3) for k in ("sac", "rainbow")}, flush=True)
