#!/usr/bin/env python3
"""Quick end-to-end check of PPOTrainer on the GPU (CartPole by default)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gymrl_amd.ppo_lunarlander import Config, PPOTrainer

env = sys.argv[1] if len(sys.argv) > 1 else "CartPole-v1"
cfg = Config()
cfg.env_name = env
cfg.num_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg.update_freq = int(sys.argv[3]) if len(sys.argv) > 3 else 128
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 15
cfg.num_minibatches = 32
cfg.num_epochs = 4
cfg.seed = None
cfg.reset_each_rollout = False
cfg.solved_reward = 1e9
tr = PPOTrainer(cfg)
for it in range(iters):
    torch.cuda.synchronize(); t0 = time.time()
    nv = tr.collect_rollout()
    torch.cuda.synchronize(); t1 = time.time()
    m = tr.update(nv)
    torch.cuda.synchronize(); t2 = time.time()
    n = cfg.num_envs * cfg.update_freq
    avg = sum(tr.episode_rewards) / max(1, len(tr.episode_rewards))
    print(f"it {it} rollout {n/(t1-t0)/1e6:.2f} Msteps/s update {t2-t1:.3f}s total {n/(t2-t0)/1e6:.2f} Msteps/s "
          f"avg_ep_ret {avg:.1f} kl {m['approx_kl']:.4f} ent {m['entropy']:.3f} clip {m['clip_frac']:.3f} vl {m['value_loss']:.3f}")
print("eval:", tr.eval(8))
