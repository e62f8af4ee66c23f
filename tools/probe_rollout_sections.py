#!/usr/bin/env python3
"""Where a vector step of the persistent rollout goes: section timers of the probe build
(`make -C gymrl_amd/csrc prof`, loaded through GYMRL_HIP_LIB), summed per workgroup, reported in us per step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GYMRL_HIP_LIB", os.path.join(ROOT, "gymrl_amd", "libgymrl_hip_prof.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gymrl_amd.ppo_lunarlander import Config, PPOTrainer  # noqa: E402

N, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = Config()
cfg.num_envs, cfg.update_freq, cfg.seed, cfg.rollout_chunk = N, T, 0, T
cfg.rollout_refill = os.environ.get("GYMRL_REFILL", "1") != "0"
sys.stdout = open(os.devnull, "w")
tr = PPOTrainer(cfg)
sys.stdout = sys.__stdout__
tr.collect_rollout()
tr.rollout_count = 0                                   # profile the same rollout whatever ran before
G = N // 16
tr._wg_ticks = torch.zeros(2 * G + 24 * G + 3 * N, dtype=torch.int64, device=tr.device)
tr.collect_rollout()
torch.cuda.synchronize()
tk = tr._wg_ticks.cpu().numpy()
busy = (tk[1:2 * G:2] - tk[0:2 * G:2]) / 100.0 / T
sec = tk[2 * G:26 * G].reshape(G, 24) / 100.0 / T
slow = int(np.argmax(busy))
names = {0: "engines+collide", 1: "constraint init", 2: "velocity sweeps (180)", 3: "store+integrate", 4: "position iterations",
         6: "  velocity sweeps, steps with a contact in the wave", 7: "  velocity sweeps, steps without", 8: "policy forward",
         9: "GAE compose + draw", 10: "env step total (load, world_step, reward, reset, store)"}
print(f"position iterations per step: wave max {tk[2 * G:26 * G].reshape(G, 24)[:, 11].mean() / T:.2f} (slowest wg {tk[2 * G:26 * G].reshape(G, 24)[slow, 11] / T:.2f}), "
      f"lane 0's own env {tk[2 * G:26 * G].reshape(G, 24)[:, 12].mean() / T:.2f}")
raw = tk[2 * G:26 * G].reshape(G, 24)
print(f"env-steps still iterating at position iteration 10: {raw[:, 5].sum() / (N * T) * 100:.2f} % of env-steps; causes (may overlap): "
      f"contact separation {raw[:, 13].sum()}, joint position error {raw[:, 14].sum()}, joint angle error {raw[:, 15].sum()} of {raw[:, 5].sum()}")
print(f"T={T}: per-WG busy us/step mean {busy.mean():.1f} max {busy.max():.1f} (wg {slow})")
print(f"{'section':58s} {'mean':>8s} {'p90':>8s} {'slowest wg':>11s}")
for k, name in names.items():
    print(f"{name:58s} {sec[:, k].mean():8.1f} {np.percentile(sec[:, k], 90):8.1f} {sec[slow, k]:11.1f}")
