"""Time one hyper-connection sub-block backward at a micro-batch of 262144 rows: the five launches vs gymrl_mhc_sub_backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402


def timed(fn, iters=20):
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    dev, D = "cuda", 128
    torch.manual_seed(0)
    h, g = torch.randn(B, 2, D, device=dev), torch.randn(B, 2, D, device=dev)
    norm_w, w = torch.empty(256, device=dev).uniform_(0.5, 1.5), torch.randn(256, 8, device=dev) * 0.3
    alpha, beta = torch.tensor([0.7, -0.4, 0.9], device=dev), torch.randn(8, device=dev) * 0.1
    W, b = torch.randn(D, D, device=dev) / 11.0, torch.randn(D, device=dev) * 0.1
    pre, post, mix, stats, read, z, _ = ops.mhc_sub_forward(h, norm_w, w, alpha, beta, W, b, 10)
    d_read = torch.empty_like(read)

    def launches():
        d_post, d_mix, d_z, _ = ops.mhc_combine_bwd(g, post, mix, z, h, act=ops.LIN_ACT["silu"], want_dh=False)
        ops.linear_bwd_input(d_z, W, None, d_read)
        d_pre, _ = ops.mhc_read_bwd(d_read, pre, h, want_dh=False)
        ops.mhc_gates_bwd(h, norm_w, w, alpha, pre, post, mix, stats, d_pre, d_post, d_mix, d_read=d_read, g_out=g)

    fused = lambda: ops.mhc_sub_backward(g, h, z, pre, post, mix, stats, norm_w, w, alpha, W)  # noqa: E731
    fwd = lambda: ops.mhc_sub_forward(h, norm_w, w, alpha, beta, W, b, 10)  # noqa: E731
    t_l, t_f, t_w = timed(launches), timed(fused), timed(fwd)
    bytes_row = 1024 + 1024 + 512 + 512 + 1024 + 4 * (2 + 2 + 4 + 9)
    print(f"rows {B}: five launches {t_l:.1f} us, one launch {t_f:.1f} us ({bytes_row * B / t_f / 1e3:.0f} GB/s of {bytes_row} B/row), "
          f"sub-block forward {t_w:.1f} us")


def phases():
    """GYMRL_HIP_LIB=gymrl_amd/libgymrl_hip_prof.so python tools/micro_sub_bwd.py --phases: shader cycles per phase of wave 0 of
    every workgroup of gymrl_mhc_sub_backward, per 16-row tile."""
    import ctypes as C
    from gymrl_amd import _lib
    B, dev, D = 262144, "cuda", 128
    torch.manual_seed(0)
    h, g = torch.randn(B, 2, D, device=dev), torch.randn(B, 2, D, device=dev)
    norm_w, w = torch.empty(256, device=dev).uniform_(0.5, 1.5), torch.randn(256, 8, device=dev) * 0.3
    alpha, beta = torch.tensor([0.7, -0.4, 0.9], device=dev), torch.randn(8, device=dev) * 0.1
    W, b = torch.randn(D, D, device=dev) / 11.0, torch.randn(D, device=dev) * 0.1
    pre, post, mix, stats, read, z, _ = ops.mhc_sub_forward(h, norm_w, w, alpha, beta, W, b, 10)
    fn = _lib.lib().gymrl_mhc_sub_bwd_prof_read
    out = (C.c_ulonglong * 8)()
    for _ in range(2):
        ops.mhc_sub_backward(g, h, z, pre, post, mix, stats, norm_w, w, alpha, W)
    torch.cuda.synchronize()
    fn(out, 1)
    n = 5
    for _ in range(n):
        ops.mhc_sub_backward(g, h, z, pre, post, mix, stats, norm_w, w, alpha, W)
    torch.cuda.synchronize()
    fn(out, 1)
    tiles = n * 256 * 16                       # wave 0 of 256 workgroups, 16 tiles each
    names = ["issue loads", "P1 (+ wait for rows)", "P2 MFMA + LDS", "d_read back, d_pre", "P3 gates", "P4", "prefetch wait", "loop"]
    tot = sum(out)
    for k, nm in enumerate(names):
        print(f"  {nm:24s} {out[k] / tiles:9.0f} cycles per tile  ({100.0 * out[k] / tot:4.1f} %)")
    print(f"  total {tot / tiles:.0f} cycles per tile")


if __name__ == "__main__":
    if "--phases" in sys.argv:
        phases()
        sys.exit(0)
    main()
