#!/usr/bin/env python3
"""PPO-full (mHC network through PyTorch) at minibatch 1024: one collect + update iteration, A/B in one process of the
GEMM path for small minibatches (fused-bias addmm on hipBLASLt's default pick vs plain GEMM on rocBLAS + bias add)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import nn as gnn  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from blas_pref import small_gemm_backend  # noqa: E402
from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer  # noqa: E402


def run(split, backend, graphs):
    gnn.SPLIT_BIAS = split
    cfg = Config()
    cfg.use_graphs = graphs
    cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed = 256, 64, 2, 1024, 1
    tr = PPOTrainer(cfg)
    with small_gemm_backend(backend):
        for it in range(4):
            if it == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            tr.collect_experience()
            adv, ret = tr.compute_advantages()
            tr.update_model(adv, ret)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 2 * 1e3


out = {"eager, addmm on hipBLASLt (before)": round(run(False, "default", False), 1),
       "eager, mm on rocBLAS + bias add": round(run(True, "rocblas", False), 1),
       "minibatch body replayed as a hipGraph (default)": round(run(True, "rocblas", True), 1)}
print(json.dumps({"ppo_full 256 envs x 64 steps, 2 epochs x 16 minibatches of 1024: ms per iteration": out}, indent=1))
