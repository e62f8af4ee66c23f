import os, sys, torch
sys.path.insert(0, '/root/repo')
from gymrl_amd import rainbow_dqn_cartpole
c = rainbow_dqn_cartpole.Config()
c.num_envs, c.memory_capacity, c.max_episodes, c.batch_size = 8192, 1 << 20, 10**9, int(sys.argv[1])
sys.stdout = open(os.devnull, "w")
tr = rainbow_dqn_cartpole.RainbowDQNTrainer(c)
tr.train(max_vector_steps=int(sys.argv[2]))
torch.cuda.synchronize()
