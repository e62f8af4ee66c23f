#!/usr/bin/env python3
"""A few launches of the split-bf16 forward (and the exact one) at B = 262,144 for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import sb
dev = torch.device("cuda:0")
B = 262144
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, 256, device=dev, generator=g)
W = torch.randn(256, 256, device=dev, generator=g) / 16
b = torch.randn(256, device=dev, generator=g)
y = torch.empty(B, 256, device=dev)
for _ in range(3):
    sb.linear_fwd_sb(x, W, b, y, act=False)
    ops.linear_fwd(x, W, b, y, act=False)
torch.cuda.synchronize()
