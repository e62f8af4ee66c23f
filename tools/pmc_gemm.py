#!/usr/bin/env python3
"""Three minibatch updates of the round-2 PPO path at bench shape (B = 262,144: ppo_net.step() = every kernel of one
minibatch) plus the library GEMMs of the same shapes, for rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ / GRBM
counters, each in its own run with --kernel-trace only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops, ppo_net
from gymrl_amd.flat import flatten_module
from gymrl_amd.ppo_lunarlander import ActorCritic
dev = torch.device("cuda:0")
B = 262144
torch.manual_seed(0)
net = ActorCritic(8, 4, 256)
flatten_module(net, dev, order=ppo_net.LAYOUT)
fu = ppo_net.FusedActorCriticUpdate(net, B)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 8, device=dev, generator=g)
act = torch.randint(0, 4, (B,), device=dev, generator=g, dtype=torch.int32)
lpo = torch.full((B,), -1.386, device=dev)
adv, ret = torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g)
mom = torch.tensor([float(B), 0.0, float(B)], dtype=torch.float64, device=dev)
parts = torch.zeros(fu.metric_blocks(B), 5, dtype=torch.float64, device=dev)
h1, w2 = torch.randn(B, 256, device=dev, generator=g), torch.randn(256, 256, device=dev, generator=g) / 16
y = torch.empty(B, 256, device=dev)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)     # evict the 256 MiB Infinity Cache between updates
for _ in range(4):
    junk.zero_()
    fu.step(x, act, lpo, adv, ret, (0.2, 3.0, 0.5, 0.01), mom, parts)
    torch.mm(h1, w2.t(), out=y)
torch.cuda.synchronize()
# provenance of the counters: which binary ran (tools/pmc_gemm_summarise.py puts it into the summary; bench.py checks it)
if os.environ.get("GYMRL_PMC_PROVENANCE"):
    import json
    from gymrl_amd import _lib
    json.dump({"libgymrl_hip_sha256": _lib.lib_sha256(), "device": torch.cuda.get_device_name(0)},
              open(os.environ["GYMRL_PMC_PROVENANCE"], "w"))
