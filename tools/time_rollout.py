#!/usr/bin/env python3
"""us per vector step of the persistent LunarLander rollout at N = 4096 (one launch per rollout), T from argv."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd.ppo_lunarlander import Config, PPOTrainer  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = Config()
cfg.num_envs, cfg.update_freq, cfg.seed, cfg.persistent_rollout, cfg.rollout_chunk = 4096, T, 0, True, T
sys.stdout = open(os.devnull, "w")
tr = PPOTrainer(cfg)
sys.stdout = sys.__stdout__
for refill in (False, True, False, True):             # A/B in one process: the in-kernel refill wave off / on
    cfg.rollout_refill = refill
    tr.rollout_count = 0                               # the SAME rollout every time (Philox counters restart)
    tr.collect_rollout()
    torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        tr.rollout_count = 0
        t0 = time.perf_counter()
        tr.collect_rollout()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / T * 1e6)
    print(f"refill={refill}: us per vector step:", [round(t, 1) for t in ts], "episodes finished per rollout:",
          int(tr.buffer.dones.sum()))
