#!/usr/bin/env python3
"""Load balance of the persistent rollout: per-workgroup busy time of one whole-rollout launch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd.ppo_lunarlander import Config, PPOTrainer
N, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = Config(); cfg.num_envs, cfg.update_freq, cfg.seed, cfg.rollout_chunk = N, T, 0, T
sys.stdout = open(os.devnull, "w"); tr = PPOTrainer(cfg); sys.stdout = sys.__stdout__
tr.collect_rollout()
tr._wg_ticks = torch.zeros(2 * (N // 16), dtype=torch.int64, device=tr.device)
tr.collect_rollout(); torch.cuda.synchronize()
tk = tr._wg_ticks.cpu().numpy().reshape(-1, 2)
busy = (tk[:, 1] - tk[:, 0]) / 100.0 / T          # us per step per workgroup
span = (tk[:, 1].max() - tk[:, 0].min()) / 100.0 / T
print(f"per-WG busy us/step: mean {busy.mean():.1f} p10 {np.percentile(busy,10):.1f} p50 {np.percentile(busy,50):.1f} p90 {np.percentile(busy,90):.1f} max {busy.max():.1f}; launch span {span:.1f} us/step; start skew {(tk[:,0].max()-tk[:,0].min())/100.0:.1f} us")
