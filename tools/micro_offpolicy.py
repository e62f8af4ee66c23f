#!/usr/bin/env python3
"""Throughput of the off-policy trainers at BASELINE configs 3 and 4 (Rainbow 8192 CartPole envs with a 2^20 PER
ring, SAC 4096 Pendulum envs), one update per vector step as in the reference loops.  Supplementary numbers:
bench.py's headline is config 2.  A/B in one process: per-launch issue vs hipGraph replay of the update, and the
library answering the round-1 path's small-M GEMMs (tools/blas_pref.py)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from blas_pref import small_gemm_backend  # noqa: E402
from gymrl_amd import rainbow_dqn_cartpole, sac_pendulum  # noqa: E402


def run(tr, steps, warm):
    sys.stdout = open(os.devnull, "w")
    tr.train(max_vector_steps=warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.env = type(tr.env)(tr.cfg.env_name, tr.cfg.num_envs, device=tr.device, seed=tr.base_seed)   # train() closed it
    tr.train(max_vector_steps=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sys.stdout = sys.__stdout__
    return dt


from gymrl_amd import nn as gnn  # noqa: E402

# name, hipGraph, GEMM library of the non-fused path, TunableOp, fused layers (csrc/lin.hip)
MODES = (("eager, library GEMM + elementwise launches (round 1)", False, "auto", False, False),
         ("graph, library GEMM + elementwise launches (round 1)", True, "auto", False, False),
         ("eager, one launch per layer and direction", False, "auto", False, True),
         ("graph, one launch per layer and direction (default config)", True, "auto", False, True))
if os.environ.get("GYMRL_MICRO_ALL"):
    MODES = (("eager hipblaslt-default", False, "default", False, False), ("graph hipblaslt-default", True, "default", False, False),
             ("graph rocblas", True, "rocblas", False, False), ("graph rocblas+tunableop", True, "rocblas", True, False)) + MODES


def main():
    out = {}
    torch.cuda.tunable.set_filename("/tmp/gymrl_tunable.csv")
    cases = ((rainbow_dqn_cartpole, "RainbowDQNTrainer", 8192, 256), (rainbow_dqn_cartpole, "RainbowDQNTrainer", 8192, 8192),
             (sac_pendulum, "SACTrainer", 4096, 128), (sac_pendulum, "SACTrainer", 4096, 4096))
    only = os.environ.get("GYMRL_MICRO_ONLY")
    for mod, cls, N, B in cases:
        if only and only not in cls:
            continue
        row = {}
        for name, graphs, backend, tune, fused in MODES:
            c = mod.Config()
            c.num_envs, c.memory_capacity, c.max_episodes, c.batch_size = N, 1 << 20, 10**9, B
            c.use_graphs = graphs
            if os.environ.get("GYMRL_NO_CHUNK"):
                c.chunk_steps = 0
            gnn.SPLIT_BIAS, gnn.FUSED_LINEAR = True, fused
            rainbow_dqn_cartpole.OVERLAP_TREE = not os.environ.get("GYMRL_NO_OVERLAP")
            tr = getattr(mod, cls)(c)
            steps = 300
            pref = ("rocblas" if B <= 4096 else "default") if backend == "auto" else backend
            with small_gemm_backend(pref, tune):       # only the non-fused (round-1) modes reach a library GEMM
                dt = run(tr, steps, 60)
            row[name] = dict(env_steps_per_s=round(N * steps / dt), ms_per_vector_step=round(dt / steps * 1e3, 3))
        out[f"{cls} N={N} cap=2^20 B={B}"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
