"""Time PPOTrainer.collect_rollout() on CartPole-v1: persistent launch vs the step-by-step loop (env-steps/s)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd.ppo_lunarlander import Config, PPOTrainer


def run(persistent, N, T, hidden, reps=5):
    cfg = Config()
    cfg.env_name = "CartPole-v1"
    cfg.num_envs, cfg.update_freq, cfg.hidden_dim, cfg.seed = N, T, hidden, 1
    cfg.persistent_rollout, cfg.rollout_chunk = persistent, 0
    tr = PPOTrainer(cfg)
    tr.collect_rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tr.collect_rollout()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dict(persistent=persistent, num_envs=N, T=T, hidden=hidden, ms_per_rollout=1e3 * dt,
                us_per_vector_step=1e6 * dt / T, env_steps_per_s=N * T / dt)


if __name__ == "__main__":
    out = [run(p, N, 128, h) for (N, h) in ((4096, 64), (4096, 256), (16384, 64)) for p in (True, False)]
    for r in out:
        print(json.dumps(r))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
