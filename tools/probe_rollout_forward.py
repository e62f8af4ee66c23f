#!/usr/bin/env python3
"""Where the persistent rollout's policy forward spends its time, per stage interval of forward_tile (probe build:
make -C gymrl_amd/csrc prof; GYMRL_HIP_LIB=gymrl_amd/libgymrl_hip_prof.so python tools/probe_rollout_forward.py)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GYMRL_HIP_LIB", os.path.join(ROOT, "gymrl_amd", "libgymrl_hip_prof.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gymrl_amd import _lib  # noqa: E402
from gymrl_amd.ppo_lunarlander import Config, PPOTrainer  # noqa: E402

cfg = Config()
cfg.num_envs, cfg.update_freq, cfg.seed = 4096, 512, 0
sys.stdout = open(os.devnull, "w")
tr = PPOTrainer(cfg)
sys.stdout = sys.__stdout__
tr.collect_rollout()
torch.cuda.synchronize()
L = _lib.lib()
out = (C.c_ulonglong * 16)()
assert L.gymrl_mlp_fwd_prof_read(out, 1) == 0
tr.rollout_count = 0
tr.collect_rollout()
torch.cuda.synchronize()
assert L.gymrl_mlp_fwd_prof_read(out, 0) == 0
n = max(1, out[15])
print("calls", n, "| us per call and stage interval:", [round(out[i] / 100.0 / n, 2) for i in range(8)], "| total", round(sum(out[:8]) / 100.0 / n, 2))
