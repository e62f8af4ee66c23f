#!/usr/bin/env python3
"""Split-bf16 GEMMs (csrc/gemm_sb.hip) beside the exact f32-MFMA kernels (csrc/gemm.hip) and the library GEMM at the
update's shapes: time per launch and error against float64."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402
import subprocess  # noqa: E402

from gymrl_amd import ops  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "tools", "probes", "libgemm_sb.so")


def _probe():
    """Build tools/probes/gemm_sb.hip (the split-bf16 probe kernel is not part of the product library)."""
    src = os.path.join(ROOT, "tools", "probes", "gemm_sb.hip")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-I", os.path.join(ROOT, "gymrl_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                               src, "-o", _SO] + [f"-DSB_ABL={os.environ['SB_ABL']}"] * ("SB_ABL" in os.environ))
    return C.CDLL(_SO)


def linear_fwd_split_bf16(x, W, b, out, act=True):
    rc = _probe().gymrl_linear_fwd_split_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()),
                                              C.c_int64(x.shape[0]), C.c_int(x.shape[1]), C.c_int(W.shape[0]), C.c_int(int(act)),
                                              C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(f"gymrl_linear_fwd_split_bf16: {rc}")
    return out


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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    torch.manual_seed(0)
    out = {}
    for N in (256, 512):
        x = torch.tanh(torch.randn(B, 256, device="cuda"))
        W = torch.randn(N, 256, device="cuda") / 16
        b = torch.randn(N, device="cuda") * 0.1
        y0, y1 = torch.empty(B, N, device="cuda"), torch.empty(B, N, device="cuda")
        sub = slice(0, 8192)
        ref = (x[sub].double() @ W.double().t() + b.double())
        ops.linear_fwd(x, W, b, y0, act=False)
        linear_fwd_split_bf16(x, W, b, y1, act=False)
        y2 = torch.addmm(b, x, W.t())
        row = {"exact_f32_mfma_us": round(timeit(lambda: ops.linear_fwd(x, W, b, y0, act=False)), 1),
               "split_bf16_us": round(timeit(lambda: linear_fwd_split_bf16(x, W, b, y1, act=False)), 1),
               "library_addmm_us": round(timeit(lambda: torch.addmm(b, x, W.t())), 1),
               "max_err_vs_f64 / max|y|": {"exact_f32_mfma": err(y0[sub], ref), "split_bf16": err(y1[sub], ref),
                                           "library_addmm": err(y2[sub], ref)},
               "split_vs_exact_max_abs": float((y0 - y1).abs().max())}
        tail = torch.empty(1000, N, device="cuda")
        linear_fwd_split_bf16(x[:1000].contiguous(), W, b, tail, act=True)
        row["ragged_1000_rows_tanh_err"] = err(tail, torch.tanh(x[:1000].double() @ W.double().t() + b.double()))
        out[f"fwd B={B} K=256 N={N}"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
