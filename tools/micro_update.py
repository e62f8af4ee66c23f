#!/usr/bin/env python3
"""One PPO minibatch update (forward, loss, backward, clip+Adam) at bench shape, A/B between variants in
the SAME process (different gpurun boxes differ by several percent).  Usage: python tools/micro_update.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops, ppo_net  # noqa: E402
from gymrl_amd.flat import FusedAdam, flatten_module  # noqa: E402
from gymrl_amd.ppo_lunarlander import ActorCritic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = 262144
    torch.manual_seed(0)
    net = ActorCritic(8, 4, 256)
    flat, grads = flatten_module(net, dev, order=ppo_net.LAYOUT)
    opt = FusedAdam(flat, grads, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    fu = ppo_net.FusedActorCriticUpdate(net, B)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 8, device=dev, generator=g)
    act = torch.randint(0, 4, (B,), device=dev, generator=g, dtype=torch.int32)
    lpo = torch.full((B,), -1.386, device=dev)
    adv, ret = torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g)
    mom = torch.tensor([float(B), 0.0, float(B)], dtype=torch.float64, device=dev)
    dl, dv = torch.empty(B, 4, device=dev), torch.empty(B, device=dev)
    parts = torch.zeros(ops.loss_blocks(B), 5, dtype=torch.float64, device=dev)

    def fused_step():
        lg, vl = fu.forward(x)
        ops.ppo_loss_fwd_bwd(lg, vl, act, lpo, adv, ret, (0.2, 3.0, 0.5, 0.01), adv_moments=mom, dlogits_out=dl,
                             dvalue_out=dv, workspace=parts)
        fu.backward(dl, dv)
        opt.step()

    def autograd_step():
        lg, vl = net(x)
        vl = vl.view(-1)
        ops.ppo_loss_fwd_bwd(lg, vl, act, lpo, adv, ret, (0.2, 3.0, 0.5, 0.01), adv_moments=mom, dlogits_out=dl,
                             dvalue_out=dv, workspace=parts)
        torch.autograd.backward([lg, vl], [dl, dv])
        opt.step()

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    out = {}
    for rep in range(3):
        fu.fused_heads_forward = True
        out.setdefault("fused_update_ms(heads_fwd_tanh)", []).append(round(timeit(fused_step), 3))
        fu.fused_heads_forward = False
        out.setdefault("fused_update_ms(tanh + 2 head GEMMs)", []).append(round(timeit(fused_step), 3))
        fu.fused_heads_forward, fu.overlap_dw = True, False
        out.setdefault("fused_update_ms(dW GEMMs on the main stream)", []).append(round(timeit(fused_step), 3))
        fu.overlap_dw = True
        fu.fused_heads_forward, fu.bias_in_gemm = True, True
        out.setdefault("fused_update_ms(bias in the GEMM epilogue)", []).append(round(timeit(fused_step), 3))
        fu.bias_in_gemm = False
    out["autograd_update_ms"] = [round(timeit(autograd_step), 3)]
    prev = torch.backends.cuda.preferred_blas_library()
    torch.backends.cuda.preferred_blas_library("cublas")          # rocBLAS for the 262 144-row GEMMs
    fu.fused_heads_forward, fu.overlap_dw, fu.bias_in_gemm = True, False, False
    out["fused_update_ms(GEMMs on rocBLAS)"] = [round(timeit(fused_step), 3) for _ in range(2)]
    torch.backends.cuda.preferred_blas_library(prev)
    if os.environ.get("GYMRL_TRY_TUNABLE"):
        import torch.cuda.tunable as tn
        tn.enable(True)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(50)
        tn.set_max_tuning_iterations(20)
        tn.set_filename("/tmp/tunableop.csv")
        fu.fused_heads_forward = True
        fused_step()
        torch.cuda.synchronize()
        tn.tuning_enable(False)
        out["fused_update_ms(tunable GEMMs)"] = [round(timeit(fused_step), 3) for _ in range(2)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
