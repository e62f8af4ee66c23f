#!/usr/bin/env python3
"""Where the split-bf16 forward (256 -> 256, B = 262,144) spends its time: timing-only ablations of ONE kernel template in the
probe build (make -C gymrl_amd/csrc prof; wrong results by design).  bit 1: no MFMAs, 2: no split / LDS parking, 4: no
activation loads in the loop, 8: no stores."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GYMRL_HIP_LIB"] = os.path.join(ROOT, "gymrl_amd", "libgymrl_hip_prof.so")
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from gymrl_amd import ops  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
import sb  # noqa: E402

dev = torch.device("cuda:0")
B = 262144
x, W, b = torch.randn(B, 256, device=dev), torch.randn(256, 256, device=dev) / 16, torch.randn(256, device=dev)
y = torch.empty(B, 256, device=dev)


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


names = {0: "product kernel", 1: "no MFMAs", 2: "no split / parking", 3: "no MFMAs, no split", 4: "no loads in the loop",
         6: "no split, no loads", 8: "no stores", 12: "no loads, no stores", 14: "MFMAs + LDS operand reads only"}
for abl, name in names.items():
    sb.lib().gymrl_gemm_sb_config(C.c_int(abl))
    print(f"{name:36s} {timeit(lambda: sb.linear_fwd_sb(x, W, b, y, act=False)):8.1f} us", flush=True)
sb.lib().gymrl_gemm_sb_config(C.c_int(0))
