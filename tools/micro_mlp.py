#!/usr/bin/env python3
"""gymrl_mlp_forward ablations (GPU): which part of the one-launch policy forward costs what.
Usage: python tools/micro_mlp.py [--N 4096]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402


def timeit(fn, iters=200, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, H = a.N, 256
    g = torch.Generator(device=dev).manual_seed(0)

    def lin(o, i):
        return torch.randn(o, i, device=dev, generator=g) / i ** 0.5, torch.zeros(o, device=dev)
    x = torch.randn(N, 8, device=dev, generator=g)
    W0, W1, Wa, Wa2, Wc, Wc2 = lin(H, 8), lin(H, H), lin(H, H), lin(4, H), lin(H, H), lin(1, H)
    o4, o1, oH = torch.empty(N, 4, device=dev), torch.empty(N, 1, device=dev), torch.empty(N, H, device=dev)

    def st(W, act, src, dst, out=None):
        return dict(W=ops.mlp_pack(W[0]), shape=tuple(W[0].shape), b=W[1], act=act, src=src, dst=dst, out=out)
    cases = {
        "full(tanh)": [st(W0, 1, -1, 0), st(W1, 1, 0, 1), st(Wa, 1, 1, 0), st(Wa2, 0, 0, -1, o4), st(Wc, 1, 1, 0), st(Wc2, 0, 0, -1, o1)],
        "full(relu)": [st(W0, 2, -1, 0), st(W1, 2, 0, 1), st(Wa, 2, 1, 0), st(Wa2, 0, 0, -1, o4), st(Wc, 2, 1, 0), st(Wc2, 0, 0, -1, o1)],
        "first_layer_only": [st(W0, 1, -1, 0), st(Wc2, 0, 0, -1, o1)],
        "L0+1wide(relu)": [st(W0, 2, -1, 0), st(W1, 2, 0, 1), st(Wc2, 0, 1, -1, o1)],
        "L0+2wide(relu)": [st(W0, 2, -1, 0), st(W1, 2, 0, 1), st(Wa, 2, 1, 0), st(Wc2, 0, 0, -1, o1)],
        "L0+3wide(relu)": [st(W0, 2, -1, 0), st(W1, 2, 0, 1), st(Wa, 2, 1, 0), st(Wc, 2, 0, 1), st(Wc2, 0, 1, -1, o1)],
        "L0+3wide(tanh)": [st(W0, 1, -1, 0), st(W1, 1, 0, 1), st(Wa, 1, 1, 0), st(Wc, 1, 0, 1), st(Wc2, 0, 1, -1, o1)],
        "L0+1wide_to_hbm": [st(W0, 2, -1, 0), st(W1, 2, 0, -1, oH)],
    }
    out = {}
    for name, stages in cases.items():
        d = ops.mlp_desc(stages)
        out[name] = round(timeit(lambda: ops.mlp_forward(x, d)), 2)
    out["empty_launch(categorical N=64)"] = round(timeit(lambda: ops.categorical_sample(o4[:64])), 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
