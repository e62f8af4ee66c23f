#!/usr/bin/env python3
"""The 128-wide weight gradient of PPO-full's training pass alone: dW [128, 128] = dY^T X over 524 288 rows
(gymrl_lin_bwd_weight -> lin_bwd_weight_big_lds_kernel + the slice reduction).  Usage: python tools/micro_dw128.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
g = torch.Generator(device=dev).manual_seed(0)
sets = [(torch.randn(B, 128, device=dev, generator=g), torch.randn(B, 128, device=dev, generator=g)) for _ in range(3)]   # 1.6 GB: no set stays in the MALL
dw, db = torch.empty(128, 128, device=dev), torch.empty(128, device=dev)
ws = ops.lin_workspace(B, 128, 128, 1, dev)
for dy, x in sets:
    ops.lin_bwd_weight(dy, None, x, dw, db, workspace=ws)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for i, (s, e) in enumerate(ev):
    dy, x = sets[i % 3]
    s.record()
    ops.lin_bwd_weight(dy, None, x, dw, db, workspace=ws)
    e.record()
torch.cuda.synchronize()
t = sorted(s.elapsed_time(e) for s, e in ev)[len(ev) // 2] * 1e-3
print(f"B = {B}: {t * 1e6:7.1f} us per call (weight-gradient launch + slice reduction), {2.0 * B * 128 * 128 / t / 1e12:6.1f} TFLOP/s f32, "
      f"{2.0 * B * 512 / t / 1e9:7.1f} GB/s of operands")
