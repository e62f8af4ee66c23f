#!/usr/bin/env python3
"""Throughput of the off-policy trainers at BASELINE configs 3 and 4 (Rainbow 8192 CartPole envs with a 2^20 PER
ring, SAC 4096 Pendulum envs), one update per vector step as in the reference loops.  Supplementary numbers:
bench.py's headline is config 2."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def main():
    out = {}
    c = rainbow_dqn_cartpole.Config()
    c.num_envs, c.memory_capacity, c.max_episodes = 8192, 1 << 20, 10**9
    for B in (256, 8192):
        c.batch_size = B
        tr = rainbow_dqn_cartpole.RainbowDQNTrainer(c)
        steps = 300
        dt = run(tr, steps, 60)
        out[f"rainbow N=8192 cap=2^20 B={B}"] = dict(env_steps_per_s=round(8192 * steps / dt), ms_per_vector_step=round(dt / steps * 1e3, 3))
    c = sac_pendulum.Config()
    c.num_envs, c.memory_capacity, c.max_episodes = 4096, 1 << 20, 10**9
    for B in (128, 4096):
        c.batch_size = B
        tr = sac_pendulum.SACTrainer(c)
        steps = 300
        dt = run(tr, steps, 60)
        out[f"sac N=4096 cap=2^20 B={B}"] = dict(env_steps_per_s=round(4096 * steps / dt), ms_per_vector_step=round(dt / steps * 1e3, 3))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
