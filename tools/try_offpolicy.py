#!/usr/bin/env python3
"""End-to-end sanity of the fused off-policy vector step: SAC on Pendulum-v1 and Rainbow on CartPole-v1 with many envs, the
running average of finished episodes every few thousand vector steps, then the reference's deterministic eval().
usage: python tools/try_offpolicy.py sac|rainbow [num_envs] [vector_steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

algo = sys.argv[1] if len(sys.argv) > 1 else "sac"
if algo == "sac":
    from gymrl_amd.sac_pendulum import Config, SACTrainer as Trainer
else:
    from gymrl_amd.rainbow_dqn_cartpole import Config, RainbowDQNTrainer as Trainer
cfg = Config()
cfg.num_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
total = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
cfg.seed, cfg.max_episodes = 0, 10 ** 9
if hasattr(cfg, "memory_capacity"):
    cfg.memory_capacity = max(cfg.memory_capacity, 1 << 20)
if "--layerwise" in sys.argv:
    cfg.fused_step = False
tr = Trainer(cfg)
print(f"{algo}: {cfg.num_envs} envs, fused step {getattr(cfg, 'fused_step', None)}, segments of {total // 8} vector steps (one update each; "
      f"every train() call restarts the envs, the networks and the replay ring carry over)")
done, t0 = 0, time.time()
for seg in range(8):
    n = total // 8
    tr.train(max_vector_steps=n)
    done += n
    torch.cuda.synchronize()
    er = list(tr.episode_rewards)
    avg = sum(er) / max(1, len(er))
    print(f"  vector step {done:7d}  env steps {done * cfg.num_envs:10d}  avg return of the last {len(er)} episodes {avg:9.2f}  "
          f"{done * cfg.num_envs / (time.time() - t0) / 1e6:.2f} M env-steps/s wall")
ev = tr.eval(8)
print("eval:", [round(float(x), 1) for x in ev])
