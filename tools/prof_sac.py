import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import sac_pendulum
c = sac_pendulum.Config()
c.num_envs, c.memory_capacity, c.max_episodes, c.batch_size = 4096, 1 << 20, 10**9, int(sys.argv[1])
sys.stdout = open(os.devnull, "w")
tr = sac_pendulum.SACTrainer(c)
tr.train(max_vector_steps=int(sys.argv[2]))
torch.cuda.synchronize()
