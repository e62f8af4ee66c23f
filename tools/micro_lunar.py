#!/usr/bin/env python3
"""lunar_step_kernel latency in isolation: fresh flight vs steady state, with/without refill."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops
from gymrl_amd.envs import VecEnv
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096

def run(prefetch, steps, warm, policy, drain=False):
    env = VecEnv("LunarLander-v3", N, device=dev, seed=1, prefetch_resets=prefetch)
    obs = env.reset(); nxt = torch.empty_like(obs)
    rew = torch.empty(N, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    acts = [torch.randint(0, 4, (N,), device=dev, generator=g, dtype=torch.int32) if policy == "random"
            else torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(64)]
    for s in range(warm):
        env.step(acts[s % 64], nxt, rew, done_out=done); obs, nxt = nxt, obs
    torch.cuda.synchronize()
    evs = []
    for s in range(steps):
        if drain:
            torch.cuda.synchronize()        # the previous step's refill has finished: the step kernel runs alone
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.step(acts[s % 64], nxt, rew, done_out=done); b.record()
        evs.append((a, b)); obs, nxt = nxt, obs
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    nd = env.ep_stats[0].item()
    env.close()
    return dict(med_us=round(ts[len(ts) // 2], 1), p10=round(ts[len(ts) // 10], 1), p90=round(ts[9 * len(ts) // 10], 1),
                episodes=nd)

print("N", N)
print("fresh flight (noop, first 40 steps, no contacts/resets):", run(False, 40, 0, "noop"))
print("steady random, inline resets:", run(False, 200, 300, "random"))
print("steady random, refill side stream:", run(True, 200, 300, "random"))
print("steady random, refill side stream, drained before each step:", run(True, 200, 300, "random", drain=True))
