#!/usr/bin/env python3
"""dW GEMM shapes of the PPO update (B = 262144 rows): torch default vs batched split-K."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

dev = torch.device("cuda:0")
B = 262144
for (K, N) in ((256, 256), (8, 256), (256, 4), (256, 1)):
    X = torch.randn(B, K, device=dev); dY = torch.randn(B, N, device=dev)
    ref = dY.t().mm(X)
    t0 = timeit(lambda: dY.t().mm(X))
    res = {"default dY^T@X": t0}
    for S in (16, 32, 64, 128, 256):
        f = lambda: torch.bmm(dY.view(S, B // S, N).transpose(1, 2), X.view(S, B // S, K)).sum(0)
        err = (f() - ref).abs().max().item() / ref.abs().max().item()
        res[f"bmm S={S}"] = (round(timeit(f), 1), f"{err:.1e}")
    f2 = lambda: torch.einsum("sbn,sbk->nk", dY.view(64, B // 64, N), X.view(64, B // 64, K))
    res["einsum S=64"] = round(timeit(f2), 1)
    res["bias dY.sum(0)"] = round(timeit(lambda: dY.sum(0)), 1)
    res["fwd X@W^T+b"] = round(timeit(lambda: torch.addmm(torch.zeros(N, device=dev), X, torch.randn(N, K, device=dev).t())), 1)
    W = torch.randn(N, K, device=dev)
    res["dX dY@W"] = round(timeit(lambda: dY.mm(W)), 1)
    print((K, N), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}, "GFLOP", 2 * B * K * N / 1e9)
