#!/usr/bin/env python3
"""Layer kernels (csrc/lin.hip) at PPO-full's skinny shapes and 262144 rows against torch's matmul: forward, input gradient and
weight gradient, microseconds per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, K, G = 262144, 256, 8
x = torch.randn(B, K, device="cuda"); wt = torch.randn(G, K, device="cuda"); dy = torch.randn(B, G, device="cuda")
y = torch.empty(B, G, device="cuda"); dwt = torch.empty(G, K, device="cuda")
ws = ops.lin_workspace(B, G, K, 1, x.device)
print("lin_fwd us", timeit(lambda: ops.lin_fwd(x, wt, None, out=y)))
print("lin_bwd_input us", timeit(lambda: ops.lin_bwd_input(dy, None, wt)))
print("lin_bwd_weight us", timeit(lambda: ops.lin_bwd_weight(dy, None, x, dwt, None, workspace=ws)))
print("torch mm fwd us", timeit(lambda: x @ wt.t()))
print("torch mm dx us", timeit(lambda: dy @ wt))
print("torch mm dw us", timeit(lambda: dy.t() @ x))
for (K2, N2) in ((256, 4), (256, 1), (8, 128)):
    x2 = torch.randn(B, K2, device="cuda"); w2 = torch.randn(N2, K2, device="cuda"); dy2 = torch.randn(B, N2, device="cuda")
    dw2 = torch.empty(N2, K2, device="cuda"); db2 = torch.empty(N2, device="cuda")
    ws2 = ops.lin_workspace(B, N2, K2, 1, x.device)
    print(K2, N2, "fwd", timeit(lambda: ops.lin_fwd(x2, w2, None)), "dx", timeit(lambda: ops.lin_bwd_input(dy2, None, w2)),
          "dw", timeit(lambda: ops.lin_bwd_weight(dy2, None, x2, dw2, db2, workspace=ws2)),
          "| torch fwd", timeit(lambda: x2 @ w2.t()), "dx", timeit(lambda: dy2 @ w2), "dw", timeit(lambda: dy2.t() @ x2))
