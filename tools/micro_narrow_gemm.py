#!/usr/bin/env python3
"""The 128- / 64-wide weight-stationary GEMMs (csrc/gemm.hip gemm_ns_kernel) at PPO-full's micro-batch (262,144 rows) against
the library's answer for the same product, A/B in one process: us per launch, TF/s and algorithmic GB/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
out = {}
for K, N in ((128, 128), (128, 256), (64, 64), (64, 128)):
    x, W, b = torch.randn(B, K, device=dev), torch.randn(N, K, device=dev) / 8, torch.randn(N, device=dev)
    y, dy, dx = torch.empty(B, N, device=dev), torch.randn(B, N, device=dev), torch.empty(B, K, device=dev)
    fl, by = 2.0 * B * K * N, 4.0 * B * (K + N)
    for name, fn in ((f"fwd {K}->{N} hip", lambda: ops.linear_fwd(x, W, b, y, act=False)),
                     (f"fwd {K}->{N} library addmm", lambda: torch.addmm(b, x, W.t(), out=y)),
                     (f"dX {N}->{K} hip", lambda: ops.linear_bwd_input(dy, W, None, dx)),
                     (f"dX {N}->{K} library mm", lambda: torch.mm(dy, W, out=dx))):
        us = timeit(fn)
        out[name] = dict(us=round(us, 1), TFLOPs=round(fl / us / 1e6, 1), GBps=round(by / us / 1e3, 1))
print(json.dumps(out, indent=1))
