#!/usr/bin/env python3
"""Runs in the BUILD container only (it imports /root/reference): the reference's own SACTrainer.train() on a numpy Pendulum-v1
(gymnasium's published step / reset arithmetic, float64 state, TimeLimit 200) behind a stub `gymnasium` module — to see how fast
the REFERENCE learns on these dynamics, as the yardstick for tools/try_offpolicy.py's curve.  usage: ref_learning_check.py [episodes] [sac|td3|ddpg] [rng seed]"""
import importlib.util
import sys
import types

import numpy as np


class _Box:
    def __init__(self, shape, high=None):
        self.shape, self.high = shape, None if high is None else np.array(high, np.float32)


class Pendulum:
    def __init__(self):
        self.observation_space, self.action_space = _Box((3,)), _Box((1,), [2.0])
        self.spec = types.SimpleNamespace(max_episode_steps=200)
        self.rng = np.random.default_rng(0)

    def _obs(self):
        th, thd = self.state
        return np.array([np.cos(th), np.sin(th), thd], np.float32)

    def reset(self, seed=None):
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        self.state = self.rng.uniform([-np.pi, -1.0], [np.pi, 1.0])
        self.t = 0
        return self._obs(), {}

    def step(self, u):
        th, thd = self.state
        u = float(np.clip(u, -2.0, 2.0)[0])
        cost = (((th + np.pi) % (2 * np.pi)) - np.pi) ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
        thd = np.clip(thd + (15.0 * np.sin(th) + 3.0 * u) * 0.05, -8.0, 8.0)
        th = th + thd * 0.05
        self.state = np.array([th, thd])
        self.t += 1
        return self._obs(), -cost, False, self.t >= 200, {}

    def close(self):
        pass


g = types.ModuleType("gymnasium")
g.make = lambda name, **kw: Pendulum()
sys.modules["gymnasium"] = g
algo = sys.argv[2] if len(sys.argv) > 2 else "sac"
spec = importlib.util.spec_from_file_location("ref_" + algo, f"/root/reference/algorithms/{algo}_pendulum.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
cfg = ref.Config()
cfg.max_episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg.seed = None
cfg.device = "cpu"
if len(sys.argv) > 3:          # the global generators the reference draws from (cfg.seed stays None: fresh starts)
    import random
    import torch
    random.seed(int(sys.argv[3])); np.random.seed(int(sys.argv[3])); torch.manual_seed(int(sys.argv[3]))
tr = {"sac": "SACTrainer", "td3": "TD3Trainer", "ddpg": "DDPGTrainer"}[algo]
tr = getattr(ref, tr)(cfg)
tr.train()
