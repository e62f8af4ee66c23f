"""DDPG — MI355X engine behind the reference's algorithms/ddpg_pendulum.py surface: Config :28-41,
Actor :44-58, Critic :62-73 (one Q network), ReplayBuffer :76-97, DDPGTrainer :100-296 (soft_update
:128-133, select_action :135-148, update :150-194, train / eval / test).

Same kernels as td3_pendulum.py; the single critic's loss is `gymrl_mse_loss` and the Bellman target
is `gymrl_sac_target` fed the one target Q twice with a zero log-prob term.
"""
import torch
import torch.nn as nn

from . import ops
from .nn import SmallLinear, frozen_parameters
from .td3_pendulum import Actor, _ActorCriticBase  # noqa: F401  (Actor is part of this module's surface)


class Config:
    def __init__(self):
        self.env_name = "Pendulum-v1"
        self.seed = None
        self.max_episodes = 500
        self.max_steps = 200
        self.batch_size = 128
        self.gamma = 0.99
        self.lr_actor = 1e-3
        self.lr_critic = 1e-3
        self.tau = 0.005
        self.noise_std = 0.1
        self.memory_capacity = 100000
        self.hidden_dim = 256
        self.device = "cuda"
        # --- vectorised-engine additions ---
        self.num_envs = 1
        self.updates_per_step = 1


class Critic(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim):
        super().__init__()
        self.fc1 = SmallLinear(state_dim + action_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc3 = SmallLinear(hidden_dim, 1)

    def forward(self, state, action):
        return self.fc3(self.fc2(self.fc1(state, action)))       # cat([state, action]) happens inside the first launch


class DDPGTrainer(_ActorCriticBase):
    def __init__(self, config):
        self._setup(config, Critic)

    def _exploration_std(self):
        return self.cfg.noise_std

    def update(self, indices=None):
        """:150-194 -> (actor_loss, critic_loss) python floats."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return 0.0, 0.0
        if indices is None and self._parity_updates is not None:
            indices = next(self._parity_updates)
        if indices is None:
            indices = self.memory.draw_indices(cfg.batch_size)
        B = self._update_body(indices)
        return -float(self._sum_a.item()) / B, float(self._sum_c.item()) / B

    def _update_body(self, indices, biases=None, alpha_bias=None):
        """Everything after the index draw; biases = device views of the (critic, actor) Adams' step scalars when the
        body runs inside / ahead of a hipGraph."""
        cfg = self.cfg
        bc, ba = biases if biases is not None else (None, None)
        states, actions, rewards, next_states, dones = self.memory.gather(indices)
        B = states.shape[0]
        with torch.no_grad():                                          # :171-174
            tq = self.critic_target(next_states, self.actor_target(next_states)).view(-1)
            y = ops.sac_target(rewards, dones, tq, tq, torch.zeros_like(rewards), self._log_alpha0, cfg.gamma)
        q = self.critic(states, actions)                               # :176-181
        self._sum_c.zero_()
        dq = ops.mse_loss(q.view(-1), y, self._sum_c)
        self._critic_sink.arm()                                        # critic_optimizer.zero_grad()
        torch.autograd.backward([q], [dq.view_as(q)])
        self._critic_sink.collect()
        self.critic_optimizer.step(bias_dev=bc, polyak=(self.critic_target_flat, cfg.tau))      # + soft update :190 in the launch
        with frozen_parameters(self.critic):                           # its gradients of this loss are never used
            qa = self.critic(states, self.actor(states))               # :183-187
        self._sum_a.zero_()
        dqa = ops.neg_mean_loss(qa.view(-1), self._sum_a)
        self._actor_sink.arm()
        torch.autograd.backward([qa], [dqa.view_as(qa)])
        self._actor_sink.collect()
        self.actor_optimizer.step(bias_dev=ba, polyak=(self.actor_target_flat, cfg.tau))        # + soft update :189
        return B

    def update_async(self):
        """update() without the host round trip, replayed as a captured hipGraph (gymrl_amd/graphs.py)."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return
        if self._graph is None:
            from .graphs import GraphedUpdate
            self._graph = GraphedUpdate(self.device, cfg.batch_size, [self.critic_optimizer, self.actor_optimizer],
                                        lambda idx, biases, ab: self._update_body(idx, biases, ab))
        self._graph(self.memory, cfg.batch_size)


if __name__ == "__main__":       # python -m gymrl_amd.ddpg_pendulum [--<Config attribute> <value> ...]  (ddpg_pendulum.py:280-296)
    from .utils.cli import run_script
    run_script(Config, DDPGTrainer)
