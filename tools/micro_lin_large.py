#!/usr/bin/env python3
"""PPO-full's wide layers and hyper-connection kernels at a 262144-row micro-batch, microseconds per launch with the GB/s of
the operands each launch has to move: csrc/lin.hip (forward, input gradient — both kernels —, weight gradient) against the
library GEMMs, and csrc/mhc.hip (gates forward / backward, combine, read, RMSNorm)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B = int(os.environ.get("GYMRL_MICRO_ROWS", 262144))
dev = "cuda"
out = {"rows": B}
for N, K in ((128, 128), (256, 128)):
    x, dy = torch.randn(B, K, device=dev), torch.randn(B, N, device=dev)
    w, b = torch.randn(N, K, device=dev) * 0.1, torch.zeros(N, device=dev)
    dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    wt = w.t().contiguous()
    ws = ops.lin_workspace(B, N, K, 1, x.device)
    mb = B * (N + K) * 4 / 1e6
    r = {"fwd": timeit(lambda: ops.lin_fwd(x, w, b)),
         "dx_input_kernel": timeit(lambda: ops.lin_bwd_input(dy, None, w)),
         "dx_forward_kernel_on_wT": timeit(lambda: ops.lin_fwd(dy, wt, None)),
         "dw": timeit(lambda: ops.lin_bwd_weight(dy, None, x, dw, db, workspace=ws)),
         "torch_fwd": timeit(lambda: torch.addmm(b, x, w.t())), "torch_dx": timeit(lambda: dy @ w),
         "torch_dw": timeit(lambda: dy.t() @ x)}
    r["operand_MB"] = mb
    r["dw_GBps"] = mb / r["dw"] * 1e3
    out[f"linear_{K}x{N}"] = r
n, D = 2, 128
h, g = torch.randn(B, n, D, device=dev), torch.randn(B, n, D, device=dev)
nw, w = torch.rand(n * D, device=dev) + 0.5, torch.randn(n * D, 8, device=dev) * 0.3
alpha, beta = torch.tensor([0.7, -0.4, 0.9], device=dev), torch.randn(8, device=dev) * 0.1
pre, post, mix, read, stats = ops.mhc_gates(h, nw, w, alpha, beta, 20, stats=True)
z = torch.randn(B, D, device=dev)
silu = ops.LIN_ACT["silu"]
d_post, d_mix, d_z, _ = ops.mhc_combine_bwd(g, post, mix, z, h, act=silu, want_dh=False)
d_pre, _ = ops.mhc_read_bwd(z, pre, h, want_dh=False)
hb = B * n * D * 4 / 1e6
m = {"gates_fwd": timeit(lambda: ops.mhc_gates(h, nw, w, alpha, beta, 20, stats=True)),
     "combine_fwd": timeit(lambda: ops.mhc_combine(post, mix, z, h, act=silu)),
     "combine_bwd": timeit(lambda: ops.mhc_combine_bwd(g, post, mix, z, h, act=silu, want_dh=False)),
     "read_bwd": timeit(lambda: ops.mhc_read_bwd(z, pre, h, want_dh=False)),
     "gates_bwd": timeit(lambda: ops.mhc_gates_bwd(h, nw, w, alpha, pre, post, mix, stats, d_pre, d_post, d_mix, d_read=z, g_out=g))}
Wl, bl = torch.randn(D, D, device=dev) * 0.1, torch.zeros(D, device=dev)
m["sub_forward_one_launch"] = timeit(lambda: ops.mhc_sub_forward(h, nw, w, alpha, beta, Wl, bl, 20))
m["sub_forward_GBps"] = hb * 3.0 / m["sub_forward_one_launch"] * 1e3      # h in; h', read, z out
m["h_MB"] = hb
m["gates_fwd_GBps"] = hb * 1.5 / m["gates_fwd"] * 1e3              # h in, read out
m["gates_bwd_GBps"] = hb * 3.5 / m["gates_bwd"] * 1e3              # h, g, d_read in, d_h out
out["mhc_2x128"] = m
for Dn in (128, 256):
    x, gy, wn = torch.randn(B, Dn, device=dev), torch.randn(B, Dn, device=dev), torch.rand(Dn, device=dev) + 0.5
    out[f"rmsnorm_{Dn}"] = {"fwd": timeit(lambda: ops.rmsnorm(x, wn, 1e-6, act=silu)),
                            "bwd": timeit(lambda: ops.rmsnorm_bwd(gy, x, wn, 1e-6, silu)), "x_MB": B * Dn * 4 / 1e6}
print(json.dumps(out))
