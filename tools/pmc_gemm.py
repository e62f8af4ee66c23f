#!/usr/bin/env python3
"""A few launches of each update-path GEMM (library and hand-written) at B = 262,144 for rocprofv3 --pmc passes:
MFMA busy cycles, wave cycles and GRBM_GUI_ACTIVE (effective clock = GUI_ACTIVE / duration)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops
dev = torch.device("cuda:0")
B = 262144
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, 256, device=dev, generator=g)
h = torch.tanh(torch.randn(B, 256, device=dev, generator=g))
dy2, dy5 = torch.randn(B, 256, device=dev, generator=g), torch.randn(B, 512, device=dev, generator=g)
W2, W5 = torch.randn(256, 256, device=dev, generator=g) / 16, torch.randn(512, 256, device=dev, generator=g) / 16
b2 = torch.randn(256, device=dev, generator=g)
y2, dx, cs = torch.empty(B, 256, device=dev), torch.empty(B, 256, device=dev), torch.empty(256, device=dev)
dW2, dW5 = torch.empty(256, 256, device=dev), torch.empty(512, 256, device=dev)
ws = ops.gemm_workspace(dev)
for _ in range(3):
    torch.mm(x, W2.t(), out=y2)
    torch.mm(dy5, W5, out=dx)
    ops.linear_fwd(x, W2, b2, y2, act=True)
    ops.linear_bwd_input(dy5, W5, h, dx)
    ops.linear_bwd_input(dy2, W2, h, dx)
    ops.linear_bwd_weight(dy2, x, dW2, ws)
    ops.linear_bwd_weight(dy5, x, dW5, ws)
torch.cuda.synchronize()
