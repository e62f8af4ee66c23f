#!/usr/bin/env python3
"""One short vectorised run of a trainer outside BASELINE's configs (dqn | td3 | ddpg | dsac | ppo_lstm), for a kernel trace:
rocprofv3 --kernel-trace --stats -- python tools/prof_others.py <algo> [vector steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
algo = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sys.stdout = open(os.devnull, "w")
if algo == "ppo_lstm":
    from gymrl_amd.ppo_lstm_lunarlander import Config, PPOTrainer
    c = Config()
    c.num_envs, c.seed = 1024, 0
    tr = PPOTrainer(c)
    for _ in range(2):
        tr.collect_experience()
        adv, ret = tr.compute_advantages()
        tr.update_model(adv, ret)
else:
    from gymrl_amd import ddpg_pendulum, dqn_cartpole, sac_cartpole, td3_pendulum
    mod, cls = {"dqn": (dqn_cartpole, "DQNTrainer"), "td3": (td3_pendulum, "TD3Trainer"), "ddpg": (ddpg_pendulum, "DDPGTrainer"),
                "dsac": (sac_cartpole, "SACTrainer")}[algo]
    c = mod.Config()
    c.num_envs, c.memory_capacity, c.max_episodes = 4096, 1 << 20, 10**9
    tr = getattr(mod, cls)(c)
    tr.train(max_vector_steps=steps)
torch.cuda.synchronize()
