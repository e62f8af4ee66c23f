#!/usr/bin/env python3
"""Ablations of the update-path GEMM kernels at B = 262,144 (variants of ONE kernel template, timed in one process; the
ablated variants compute wrong results by design).  Output committed as profiles/r02_gemm_ablation.txt.
gymrl_gemm_config key 4 (weight gradient): 0 product kernel, 1 no operand loads inside the loop, 2 MFMAs only.
key 5 (forward, shared.2 + tanh) bit mask: 1 no A loads inside the loop, 2 no LDS reads, 4 no epilogue stages (VALU + store),
8 epilogue VALU kept but no stores."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops
from tools.micro_gemm import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B = 262144
x = torch.randn(B, 256, device=dev, generator=g); dy2 = torch.randn(B, 256, device=dev, generator=g)
dW2 = torch.empty(256, 256, device=dev); ws = ops.gemm_workspace(dev)
W2 = torch.randn(256, 256, device=dev, generator=g) / 16; b2 = torch.randn(256, device=dev, generator=g); y2 = torch.empty(B, 256, device=dev)
f = 2.0 * B * 65536
timeit(lambda: ops.linear_fwd(x, W2, b2, y2, act=True), iters=30)          # clocks up
print(f"rows {B}; 34.4 GFLOP per launch; f32 MFMA peak 157.3 TF/s at 2.4 GHz (the chip holds ~2.19 GHz under this load: 143.5)")
names = {0: "product kernel (+ its reduce launch)", 1: "no operand loads inside the loop", 2: "MFMAs only"}
for abl in (0, 1, 2):
    ops.gemm_config(4, abl)
    t = min(timeit(lambda: ops.linear_bwd_weight(dy2, x, dW2, ws)) for _ in range(3))
    print(f"weight gradient dW 256   abl={abl:2d} {names[abl]:44s} {t:8.1f} us {f/t*1e-6:6.1f} TF/s", flush=True)
ops.gemm_config(4, 0)
names = {0: "product kernel", 8: "epilogue VALU kept, no stores", 4: "no epilogue (VALU + stores)", 12: "no epilogue", 1: "no A loads in the loop",
         2: "no LDS reads in the loop", 7: "MFMAs only"}
for abl in (0, 8, 4, 1, 2, 7):
    ops.gemm_config(5, abl)
    t = min(timeit(lambda: ops.linear_fwd(x, W2, b2, y2, act=True)) for _ in range(3))
    print(f"forward 256 + bias + tanh abl={abl:2d} {names[abl]:44s} {t:8.1f} us {f/t*1e-6:6.1f} TF/s", flush=True)
ops.gemm_config(5, 0)
t = min(timeit(lambda: torch.mm(x, W2.t(), out=y2)) for _ in range(3))
print(f"library mm 256 (hipBLASLt, no epilogue)                                          {t:8.1f} us {f/t*1e-6:6.1f} TF/s")
