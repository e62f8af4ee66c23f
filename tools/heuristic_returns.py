#!/usr/bin/env python3
"""The return distribution of gymnasium's documented LunarLander heuristic controller on the HIP stepper at 4096 envs: the
statistic tests/test_oracle_envs.py checks on the CPU restatement (mean return > 150, >= 70 % landings), measured on the
product kernel at the benchmark's env count — the env's behavioural pin in the absence of an installable gymnasium/Box2D
(SURVEY.md 8c.2).  First episode of every env; the controller runs on the device in torch (same rule as
tests/test_hip_parity.py::_lander_heuristic).  Prints a JSON summary + histogram.  usage: python tools/heuristic_returns.py [n_envs]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import _lib, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
seed = 7
state = ops.env_state(ops.LUNARLANDER, n, dev)
obs = torch.empty(n, 8, device=dev)
rew = torch.empty(n, device=dev)
term, trunc, done = (torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(3))
ep_ret = torch.zeros(n, device=dev)
ep_len = torch.zeros(n, dtype=torch.int32, device=dev)
ops.env_reset(ops.LUNARLANDER, state, n, seed, 0, obs)


def heuristic(s):
    angle_targ = (s[:, 0] * 0.5 + s[:, 2] * 1.0).clamp(-0.4, 0.4)
    hover_targ = 0.55 * s[:, 0].abs()
    angle_todo = (angle_targ - s[:, 4]) * 0.5 - s[:, 5] * 1.0
    hover_todo = (hover_targ - s[:, 1]) * 0.5 - s[:, 3] * 0.5
    legs = (s[:, 6] > 0) | (s[:, 7] > 0)
    angle_todo = torch.where(legs, torch.zeros_like(angle_todo), angle_todo)
    hover_todo = torch.where(legs, -s[:, 3] * 0.5, hover_todo)
    main = (hover_todo > angle_todo.abs()) & (hover_todo > 0.05)
    a = torch.zeros(s.shape[0], dtype=torch.int32, device=s.device)
    a = torch.where(main, torch.full_like(a, 2), a)
    a = torch.where(~main & (angle_todo < -0.05), torch.full_like(a, 3), a)
    a = torch.where(~main & (angle_todo > 0.05), torch.full_like(a, 1), a)
    return a.contiguous()


first_ret = torch.full((n,), float("nan"), device=dev)
first_len = torch.zeros(n, dtype=torch.int32, device=dev)
landed = torch.zeros(n, dtype=torch.bool, device=dev)
timeout = torch.zeros(n, dtype=torch.bool, device=dev)
for t in range(1001):
    act = heuristic(obs)
    ops.env_step(ops.LUNARLANDER, state, n, seed, 0, act, obs, rew, term, trunc, done_out=done, ep_ret_out=ep_ret, ep_len_out=ep_len)
    new = done.bool() & torch.isnan(first_ret)
    first_ret = torch.where(new, ep_ret, first_ret)
    first_len = torch.where(new, ep_len, first_len)
    landed |= new & (rew == 100.0)
    timeout |= new & trunc.bool()
    if not torch.isnan(first_ret).any():
        break
r = first_ret.cpu().numpy().astype(np.float64)
assert not np.isnan(r).any(), "every env finishes its first episode within the 1000-step time limit"
edges = list(range(-400, 351, 50))
hist, _ = np.histogram(r, bins=edges)
out = {"what": "first-episode returns of gymnasium's heuristic LunarLander controller on the HIP stepper (gymrl_env_step)",
       "n_envs": n, "seed": seed, "library_sha256": _lib.lib_sha256(), "device": torch.cuda.get_device_name(0),
       "mean_return": float(r.mean()), "median_return": float(np.median(r)), "std_return": float(r.std()),
       "min_return": float(r.min()), "max_return": float(r.max()),
       "landed_asleep_fraction": float(landed.float().mean()), "time_limit_fraction": float(timeout.float().mean()),
       "mean_episode_length": float(first_len.float().mean()),
       "fraction_above_200": float((r >= 200).mean()), "fraction_below_0": float((r < 0).mean()),
       "histogram": {"bin_edges": edges, "counts": hist.tolist()},
       "oracle_statistic_for_comparison": "tests/test_oracle_envs.py: 96 envs on the CPU restatement, mean > 150 and >= 70 % landings required (measured 235, 92 %)"}
print(json.dumps(out, indent=1))
assert out["mean_return"] > 150 and out["landed_asleep_fraction"] >= 0.7
