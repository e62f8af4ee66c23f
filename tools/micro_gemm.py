#!/usr/bin/env python3
"""The update-path GEMMs at bench shape (B = 262,144): gymrl_linear_fwd / _bwd_input / _bwd_weight against the library GEMM
of the same shape, A/B in ONE process.  With the probe build (make -C gymrl_amd/csrc prof; GYMRL_HIP_LIB=.../libgymrl_hip_prof.so)
also the 2- vs 4-slice variants of the 256-wide kernels.  Prints TF/s per kernel and writes gpurun_out/micro_gemm.json.
Usage: python tools/micro_gemm.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, 256, device=dev, generator=g)
    h = torch.tanh(torch.randn(B, 256, device=dev, generator=g))
    dy2, dy5 = torch.randn(B, 256, device=dev, generator=g), torch.randn(B, 512, device=dev, generator=g)
    W2, W5 = torch.randn(256, 256, device=dev, generator=g) / 16, torch.randn(512, 256, device=dev, generator=g) / 16
    b2, b5 = torch.randn(256, device=dev, generator=g), torch.randn(512, device=dev, generator=g)
    y2, y5 = torch.empty(B, 256, device=dev), torch.empty(B, 512, device=dev)
    dx, cs = torch.empty(B, 256, device=dev), torch.empty(256, device=dev)
    dW2, dW5 = torch.empty(256, 256, device=dev), torch.empty(512, 256, device=dev)
    ws = ops.gemm_workspace(dev)
    out = {"rows": B}

    def rec(name, us, flop):
        out[name] = {"us": round(us, 1), "tflops": round(flop / us * 1e-6, 1), "frac_f32_mfma_peak": round(flop / us * 1e-6 / 157.3, 3)}
        print(f"{name:42s} {us:9.1f} us  {flop / us * 1e-6:7.1f} TF/s", flush=True)

    f2, f5 = 2.0 * B * 256 * 256, 2.0 * B * 512 * 256
    rec("lib fwd 256 (mm only)", timeit(lambda: torch.mm(x, W2.t(), out=y2)), f2)
    rec("lib fwd 512 (mm only)", timeit(lambda: torch.mm(x, W5.t(), out=y5)), f5)
    rec("lib dX 512 (mm only)", timeit(lambda: torch.mm(dy5, W5, out=dx)), f5)
    rec("lib dX 256 (mm only)", timeit(lambda: torch.mm(dy2, W2, out=dx)), f2)
    prof = hasattr(ops.lib(), "gymrl_gemm_config")
    for slices in ((2, 4, 2, 4) if prof else (None,)):
        tag = "" if slices is None else f" [{slices} slices]"
        if slices is not None:
            ops.gemm_config(6, slices)
        rec("hip fwd 256 tanh" + tag, timeit(lambda: ops.linear_fwd(x, W2, b2, y2, act=True)), f2)
        rec("hip fwd 256 bias only" + tag, timeit(lambda: ops.linear_fwd(x, W2, b2, y2, act=False)), f2)
        rec("hip dX 256 tanh'" + tag, timeit(lambda: ops.linear_bwd_input(dy2, W2, h, dx)), f2)
    if prof:
        ops.gemm_config(6, 2)
    rec("hip fwd 512 bias only", timeit(lambda: ops.linear_fwd(x, W5, b5, y5, act=False)), f5)
    rec("hip dX 512 tanh'", timeit(lambda: ops.linear_bwd_input(dy5, W5, h, dx)), f5)
    rec("hip dW 256 + db", timeit(lambda: ops.linear_bwd_weight(dy2, x, dW2, ws, cs)), f2)
    rec("hip dW 512", timeit(lambda: ops.linear_bwd_weight(dy5, x, dW5, ws, None)), f5)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/micro_gemm.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
