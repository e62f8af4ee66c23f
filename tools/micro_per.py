#!/usr/bin/env python3
"""The sum-tree update alone, per kernel: a vector step's 8192 consecutive new rows (ordered runs) and the priorities of a
sampled batch of 256 (unordered, duplicates possible) on a 2^20-leaf tree.  Usage: python tools/micro_per.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
cap = 1 << 20
tree = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=50):
    """us per call, the calls issued eagerly back to back: at these sizes the HOST's ~15 us per launch (ctypes marshalling +
    hipLaunchKernel) is what this measures, not the kernels."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def timeit_graph(fn, per_graph=20, replays=10):
    """us per call with `per_graph` calls captured into ONE hipGraph and replayed — how the trainers issue these launches
    (gymrl_amd/graphs.py): the device's time per call plus the graph executor's per-node cost, no host in the loop."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (replays * per_graph) * 1e3


for B, kind in ((8192, "rows"), (256, "idx"), (256, "idx_dup"), (512, "idx"), (1024, "idx")):
    ws = ops.per_workspace(B, dev)
    prio = torch.rand(B, dtype=torch.float64, device=dev, generator=g) + 0.1
    if kind == "rows":
        fn = lambda: ops.per_update(tree, cap, B, ws, idx_start=12345 * 8, prio=prio)
    else:
        idx = torch.randint(0, 4096 if kind == "idx_dup" else cap, (B,), device=dev, generator=g, dtype=torch.int32)
        if kind == "idx_dup":
            idx[::3] = idx[0]
        fn = lambda: ops.per_update(tree, cap, B, ws, idx=idx, prio=prio)
    print(f"B = {B:5d} {kind:8s}: {timeit_graph(fn):7.1f} us per update replayed from a hipGraph ({timeit(fn):6.1f} issued eagerly: host-bound)")

# update_priorities as the Rainbow trainer issues it: straight from the TD errors, the next store's priority_max in the same
# ONE launch (gymrl_per_update_td with a ticket: depth blocks | maximum blocks, the last block writes the leaves)
ws = ops.per_workspace(8192, dev)
mx = torch.zeros(1, dtype=torch.float64, device=dev)
ticket = torch.zeros(1, dtype=torch.int32, device=dev)
for B in (128, 256, 512):
    idx = torch.randint(0, cap, (B,), device=dev, generator=g, dtype=torch.int32)
    td = torch.randn(B, device=dev, generator=g)
    fn = lambda: ops.per_update_td(tree, cap, idx, td, 0.6, 0.01, ws, max_out=mx, ticket=ticket)
    print(f"B = {B:5d} td+max  : {timeit_graph(fn):7.1f} us per update_td replayed from a hipGraph (one launch, leaves' maximum included; "
          f"{timeit(fn):6.1f} issued eagerly)")
