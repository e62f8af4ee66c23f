#!/usr/bin/env python3
"""PPO-full's head layers alone at a micro-batch's size: forward y [B, 256] = x [B, 128] W^T + b (gemm_ns_kernel<128, 4, false, 0, 256>)
and input gradient dx [B, 128] = dy [B, 256] W (gemm_ws_kernel<256, 4, 2, true, 2, 128>).  Usage: python tools/micro_ns.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(B, 128, device=dev, generator=g) for _ in range(3)]
dys = [torch.randn(B, 256, device=dev, generator=g) for _ in range(3)]
W, b = torch.randn(256, 128, device=dev, generator=g) * 0.05, torch.randn(256, device=dev, generator=g)
y, dx = torch.empty(B, 256, device=dev), torch.empty(B, 128, device=dev)


def timed(fn, n=30):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i, (s, e) in enumerate(ev):
        s.record()
        fn(i)
        e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[n // 2] * 1e-3


fl = 2.0 * B * 128 * 256
t = timed(lambda i: ops.linear_fwd(xs[i % 3], W, b, y, act=False))
print(f"forward        : {t * 1e6:7.1f} us, {fl / t / 1e12:6.1f} TFLOP/s f32, {B * 1536.0 / t / 1e9:7.1f} GB/s")
t = timed(lambda i: ops.linear_bwd_input(dys[i % 3], W, None, dx))
print(f"input gradient : {t * 1e6:7.1f} us, {fl / t / 1e12:6.1f} TFLOP/s f32, {B * 1536.0 / t / 1e9:7.1f} GB/s")
