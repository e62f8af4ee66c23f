#!/usr/bin/env python3
"""Where PPO-full's persistent rollout spends its policy forward (mhc_policy_device.hpp policy_tile), per phase (probe build:
make -C gymrl_amd/csrc prof; GYMRL_HIP_LIB=gymrl_amd/libgymrl_hip_prof.so python tools/probe_mhc_policy.py)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GYMRL_HIP_LIB", os.path.join(ROOT, "gymrl_amd", "libgymrl_hip_prof.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gymrl_amd import _lib  # noqa: E402
from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer  # noqa: E402

cfg = Config()
cfg.num_envs, cfg.update_freq, cfg.seed = 4096, 256, 0
if len(sys.argv) > 1:
    cfg.mhc_sk_it = int(sys.argv[1])          # (how much of the gates is the Sinkhorn loop: run with 0)
sys.stdout = open(os.devnull, "w")
tr = PPOTrainer(cfg)
sys.stdout = sys.__stdout__
tr.collect_experience()
torch.cuda.synchronize()
L = _lib.lib()
out = (C.c_ulonglong * 16)()
assert L.gymrl_mhc_policy_prof_read(out, 1) == 0
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
tr.collect_experience()
t1.record()
torch.cuda.synchronize()
assert L.gymrl_mhc_policy_prof_read(out, 0) == 0
n = max(1, out[15])
names = ("input projection", "gates + read (x n_sub)", "Linear + SiLU (x n_sub)", "combine (x n_sub)", "final norm", "heads", "tail")
print(f"calls {n}; rollout {t0.elapsed_time(t1) * 1000 / cfg.update_freq:.1f} us per vector step")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {out[i] / 100.0 / n:6.2f} us per call")
print(f"  total {sum(out[:7]) / 100.0 / n + sum(out[7:11]) / 100.0 / n:.2f} us")
print("  inside gates + read: accumulate, reductions, transcendentals + Sinkhorn, read + LDS store, (barrier = the first line):", [round(out[i] / 100.0 / n, 2) for i in range(7, 11)])
