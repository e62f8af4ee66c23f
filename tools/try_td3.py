"""Learning curve of the TD3 / DDPG trainers on the GPU box:  python tools/try_td3.py <num_envs> <vector steps> [seed] [td3|ddpg]
(the reference's own curve on the same dynamics: tools/ref_learning_check.py, build container only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
algo = sys.argv[4] if len(sys.argv) > 4 else "td3"
if algo == "td3":
    from gymrl_amd.td3_pendulum import Config, TD3Trainer as Trainer
else:
    from gymrl_amd.ddpg_pendulum import Config, DDPGTrainer as Trainer
cfg = Config()
cfg.num_envs, cfg.seed, cfg.max_episodes = int(sys.argv[1]), int(sys.argv[3]) if len(sys.argv) > 3 else 0, 10 ** 9
tr = Trainer(cfg)
SEG = 10
for seg in range(SEG):
    tr.train(max_vector_steps=int(sys.argv[2]) // SEG)
    er = list(tr.episode_rewards)
    print(f"{algo} seed {cfg.seed} segment {seg}: episodes so far (window 100) {len(er)}, mean of last 10 {sum(er[-10:]) / max(1, len(er[-10:])):.0f}", flush=True)
print("eval", [round(float(x), 1) for x in tr.eval(8)])
