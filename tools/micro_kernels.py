#!/usr/bin/env python3
"""Kernel micro-benchmarks (GPU): GAE variants, PPO loss, Adam at BASELINE config-2
sizes.  Prints achieved GB/s on ALGORITHMIC bytes and the fraction of the 8 TB/s
HBM3E peak.  Usage: python tools/micro_kernels.py [--T 2048 --N 4096]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops  # noqa: E402

PEAK = 8.0e12


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=2048)
    ap.add_argument("--N", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    T, N = a.T, a.N
    g = torch.Generator(device=dev).manual_seed(1)
    rew = torch.randn(T, N, device=dev, generator=g)
    val = torch.randn(T, N, device=dev, generator=g)
    done = (torch.rand(T, N, device=dev, generator=g) < 1 / 300).to(torch.uint8)
    nv = torch.randn(N, device=dev, generator=g)
    adv, ret = torch.empty_like(rew), torch.empty_like(rew)
    ws = ops.gae_workspace(T, N, dev)
    mom = torch.zeros(3, dtype=torch.float64, device=dev)
    out = {}
    for variant in (0, 1, 2):      # 2 reuses the chunk maps variant 1 just left in the workspace
        dt = timeit(lambda: ops.gae(rew, val, done, nv, 0.99, 0.95, adv, ret, mom, variant, ws))
        gb = 17.0 * T * N / dt
        out[f"gae_v{variant}"] = dict(us=dt * 1e6, GBps=gb / 1e9, frac=gb / PEAK)
    # copy bandwidth reference (read+write)
    src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    dt = timeit(lambda: dst.copy_(src))
    out["copy_256MiB"] = dict(us=dt * 1e6, GBps=2 * src.numel() / dt / 1e9, frac=2 * src.numel() / dt / PEAK)
    # loss
    B = T * N
    logits = torch.randn(B, 4, device=dev, generator=g)
    v = torch.randn(B, device=dev, generator=g)
    act = torch.randint(0, 4, (B,), device=dev, generator=g, dtype=torch.int32)
    lpo = torch.log_softmax(logits, -1).gather(1, act.long()[:, None]).squeeze(1) + 0.1 * torch.randn(B, device=dev, generator=g)
    dl, dv = torch.empty_like(logits), torch.empty_like(v)
    met = torch.zeros(5, dtype=torch.float64, device=dev)
    cfg = (0.2, 3.0, 0.5, 0.01)
    a1, r1 = adv.view(-1), ret.view(-1)
    dt = timeit(lambda: ops.ppo_loss_fwd_bwd(logits, v, act, lpo, a1, r1, cfg, None, mom, dl, dv, met))
    gb = 56.0 * B / dt
    out["ppo_loss"] = dict(us=dt * 1e6, GBps=gb / 1e9, frac=gb / PEAK)
    idx = torch.randperm(B, device=dev, generator=g).to(torch.int32)
    dt = timeit(lambda: ops.ppo_loss_fwd_bwd(logits, v, act, lpo, a1, r1, cfg, idx, mom, dl, dv, met))
    out["ppo_loss_gather"] = dict(us=dt * 1e6, GBps=60.0 * B / dt / 1e9, frac=60.0 * B / dt / PEAK)
    # adam on the 200,965-param PPO model and on a large buffer
    for n in (200965, 1 << 26):
        p, gr, m, vv = (torch.randn(n, device=dev, generator=g) for _ in range(4))
        vv.abs_()
        sq = torch.zeros(1, dtype=torch.float64, device=dev)
        rws = ops.reduce_workspace(dev)

        def step():
            ops.sqnorm(gr, sq, rws)
            ops.adam_step(p, gr, m, vv, 3e-4, 0.9, 0.999, 1e-5, 1, max_grad_norm=0.5, sqnorm_buf=sq, zero_grad=True)
        dt = timeit(step)
        out[f"adam_{n}"] = dict(us=dt * 1e6, GBps=36.0 * n / dt / 1e9, frac=36.0 * n / dt / PEAK)
    # rollout policy forward: one gymrl_mlp_forward launch vs the per-layer torch path
    from gymrl_amd.flat import flatten_module
    from gymrl_amd.ppo_lunarlander import ActorCritic
    net = ActorCritic(8, 4, 256)
    flatten_module(net, dev)
    obs = torch.randn(N, 8, device=dev, generator=g)
    flops = 2.0 * N * (8 * 256 + 3 * 256 * 256 + 256 * 5)
    net.refresh_act()
    dt = timeit(lambda: net.act_forward(obs, refresh=False), iters=50)
    out["mlp_forward_fused"] = dict(us=dt * 1e6, TFLOPs=flops / dt / 1e12, frac_f32_mfma=flops / dt / 157.3e12)
    with torch.no_grad():
        dt = timeit(lambda: net(obs), iters=50)
    out["mlp_forward_torch"] = dict(us=dt * 1e6, TFLOPs=flops / dt / 1e12)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
