#!/usr/bin/env python3
"""collect_rollout() per vector step: persistent launches (gymrl_rollout_lunar) vs step-by-step, for several
env counts, in one process.  Usage: python tools/micro_rollout.py [T]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd.ppo_lunarlander import Config, PPOTrainer  # noqa: E402


def run(N, T, persistent, chunk):
    cfg = Config()
    cfg.num_envs, cfg.update_freq, cfg.seed, cfg.persistent_rollout, cfg.rollout_chunk = N, T, 0, persistent, chunk
    sys.stdout = open(os.devnull, "w")
    tr = PPOTrainer(cfg)
    sys.stdout = sys.__stdout__
    tr.collect_rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        tr.collect_rollout()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * T) * 1e6


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    out = {}
    for N in (16, 256, 4096):
        out[f"N={N} stepwise"] = round(run(N, T, False, T), 1)
        for chunk in (64, 256, T):
            out[f"N={N} persistent chunk={chunk}"] = round(run(N, T, True, chunk), 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
