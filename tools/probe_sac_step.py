#!/usr/bin/env python3
"""Timeline of ONE gymrl_sac_step launch (probe build: make -C gymrl_amd/csrc prof;
GYMRL_HIP_LIB=gymrl_amd/libgymrl_hip_prof.so python tools/probe_sac_step.py): the 100 MHz stamps of the first block of every
phase, as microseconds after the acting phase's start."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import _lib  # noqa: E402
from gymrl_amd.sac_pendulum import Config, SACTrainer  # noqa: E402

cfg = Config()
cfg.num_envs, cfg.seed, cfg.max_episodes, cfg.memory_capacity = 4096, 0, 10 ** 9, 1 << 20
tr = SACTrainer(cfg)
tr.train(max_vector_steps=64)
torch.cuda.synchronize()
L = _lib.lib()
out = (C.c_longlong * 128)()
assert L.gymrl_step_prof_read(out) == 0
g = [[out[k * 32 + i] for i in range(32)] for k in range(4)]
t0 = g[2][0]
us = lambda v: round((v - t0) / 100.0, 1) if v else None  # noqa: E731
print("acting            :", [us(v) for v in g[2][:6]])
print("P1 target chain 1 :", [us(v) for v in g[3][:10]])
print("P1 critic chain 1 :", [us(v) for v in g[0][:9]])
print("P2 block 0 (start, counter met, done):", [us(v) for v in g[2][16:19]])
print("P3 workgroup 0    :", [us(v) for v in g[1][:12]])
print("P4 block 0 (start, counter met, done):", [us(v) for v in g[2][19:22]])
