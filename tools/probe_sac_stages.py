#!/usr/bin/env python3
"""Where a fused SAC step's row kernels spend their time: 100 MHz wall-clock stamps at every stage boundary of workgroup 0
(probe build: make -C gymrl_amd/csrc prof; GYMRL_HIP_LIB=gymrl_amd/libgymrl_hip_prof.so python tools/probe_sac_stages.py)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import _lib  # noqa: E402
from gymrl_amd.sac_pendulum import Config, SACTrainer  # noqa: E402

cfg = Config()
cfg.num_envs, cfg.seed, cfg.max_episodes, cfg.memory_capacity, cfg.use_graphs = 4096, 0, 10 ** 9, 1 << 20, False
tr = SACTrainer(cfg)
tr.train(max_vector_steps=24)
torch.cuda.synchronize()
L = _lib.lib()
out = (C.c_longlong * 128)()
assert L.gymrl_step_prof_read(out) == 0
names = ("P1 critic chain (draw+gather | fc1 x2 | fc2 x2 | q heads | wait for y | loss | bwd3 | bwd2)",
         "P3 workgroup 0 (load s, sample, actor slabs | Q1.fc1 | Q1.fc2 | Q1.fc3 | Q exchange + loss | bwd3 | bwd2 | wait for Q2's dZ1 + d action | sample bwd | heads bwd | fc2 bwd)",
         "acting (load | fc1 | fc2 | heads | sample+env+row)",
         "P1 target chain (draw+gather | a.fc1 | a.fc2 | heads | sample | tgt fc1 x2 | tgt fc2 x2 | tgt fc3 x2 | y)")
for k in range(4):
    st = [out[k * 32 + i] for i in range(32)]
    n = max(i for i, v in enumerate(st) if v) + 1
    d = [st[i + 1] - st[i] for i in range(n - 1)]
    print(names[k])
    print("  us:", [round(x / 100.0, 1) for x in d], " total", (st[n - 1] - st[0]) / 100.0)
