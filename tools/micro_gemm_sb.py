#!/usr/bin/env python3
"""The opt-in split-bf16 GEMMs (csrc/gemm_sb.hip) beside the exact f32-MFMA kernels (csrc/gemm.hip) and the library at the
update's shapes (B = 262,144): us per launch and error against float64.  Usage: python tools/micro_gemm_sb.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import sb  # noqa: E402  (tools/probes/sb.py: the probe library's wrappers)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def err(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max())


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    g = torch.Generator(device=dev).manual_seed(0)
    out = {"rows": B}
    x = torch.randn(B, 256, device=dev, generator=g)
    h = torch.tanh(torch.randn(B, 256, device=dev, generator=g))
    for N in (256, 512):
        W = torch.randn(N, 256, device=dev, generator=g) / 16
        b = torch.randn(N, device=dev, generator=g)
        y, y2 = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev)
        ref = (x.double() @ W.double().t() + b.double())
        fl = 2.0 * B * 256 * N
        rows = {}
        rows["exact"] = (timeit(lambda: ops.linear_fwd(x, W, b, y, act=False)), err(y, ref))
        rows["split_bf16"] = (timeit(lambda: sb.linear_fwd_sb(x, W, b, y2, act=False)), err(y2, ref))
        rows["library"] = (timeit(lambda: torch.addmm(b, x, W.t(), out=y)), err(y, ref))
        # round 4: the activations split by their producer (three bf16 planes, 6 B per element in); the producer's share is
        # what writing 1.5 x the bytes costs an HBM-bound layer kernel — measured here as the stand-alone split's time
        planes = sb.split_planes(x)
        y3 = torch.empty(B, N, device=dev)
        rows["split_bf16_producer_planes"] = (timeit(lambda: sb.linear_fwd_sb_planes(planes, W, b, y3, act=False)), err(y3, ref))
        rows["(stand-alone split of x into planes)"] = (timeit(lambda: sb.split_planes(x, planes)), 0.0)
        del planes, y3
        del ref
        out[f"fwd 256->{N}"] = {k: dict(us=round(u, 1), TFLOPs=round(fl / u / 1e6, 1), err_vs_f64=float(f"{e:.3g}")) for k, (u, e) in rows.items()}
        dy = torch.randn(B, N, device=dev, generator=g)
        dx, dx2 = torch.empty(B, 256, device=dev), torch.empty(B, 256, device=dev)
        ref = (dy.double() @ W.double()) * (1 - h.double() ** 2)
        rows = {}
        rows["exact"] = (timeit(lambda: ops.linear_bwd_input(dy, W, h, dx)), err(dx, ref))
        rows["split_bf16"] = (timeit(lambda: sb.linear_bwd_input_sb(dy, W, h, dx2)), err(dx2, ref))
        del ref
        out[f"dX {N}->256 tanh'"] = {k: dict(us=round(u, 1), TFLOPs=round(fl / u / 1e6, 1), err_vs_f64=float(f"{e:.3g}")) for k, (u, e) in rows.items()}
        if hasattr(ops, "linear_bwd_weight_sb") and hasattr(ops.lib(), "gymrl_linear_bwd_weight_sb"):
            ws = ops.gemm_workspace(dev)
            dW, dW2, db, db2 = (torch.empty(N, 256, device=dev), torch.empty(N, 256, device=dev), torch.empty(N, device=dev),
                                torch.empty(N, device=dev))
            ref = dy.double().t() @ x.double()
            rows = {}
            rows["exact"] = (timeit(lambda: ops.linear_bwd_weight(dy, x, dW, ws, db)), err(dW, ref))
            rows["split_bf16"] = (timeit(lambda: ops.linear_bwd_weight_sb(dy, x, dW2, ws, db2)), err(dW2, ref))
            out[f"dW {N}x256"] = {k: dict(us=round(u, 1), TFLOPs=round(fl / u / 1e6, 1), err_vs_f64=float(f"{e:.3g}")) for k, (u, e) in rows.items()}
            out[f"db {N} max abs diff exact vs split"] = float((db - db2).abs().max())
            del ref
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
