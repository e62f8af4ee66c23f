"""Discrete SAC (expectation over actions, twin critics with separate optimisers, float32 temperature) —
MI355X engine behind the reference's algorithms/sac_cartpole.py surface: Config :28-43, ReplayBuffer :46-67,
Actor :70-80 (softmax output), Critic :83-93 (Q per action), SACTrainer :96-329 (select_action :127-138,
soft_update :140-145, update :148-227, train / eval / test).

Underneath: CartPole instances step on the GPU; replay ring, categorical draw, soft-Bellman target, both
critic losses, the actor loss's forward + dL/dprobs, the float32 log_alpha Adam step, three fused Adam steps
and the Polyak updates are HIP kernels behind the C-ABI; Linear layers and the softmax run through PyTorch-ROCm.
"""
import copy
from collections import deque

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .dqn_cartpole import ReplayBuffer
from .envs import EpisodeTracker, VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .nn import SmallLinear
from .utils import scalar


class Config:
    def __init__(self):
        self.env_name = "CartPole-v1"
        self.seed = None
        self.max_episodes = 500
        self.max_steps = 2000
        self.batch_size = 128
        self.gamma = 0.9
        self.tau = 0.005
        self.lr_actor = 2e-4
        self.lr_critic = 1e-3
        self.lr_alpha = 1e-3
        self.memory_capacity = 10000
        self.hidden_dim = 256
        self.target_entropy = -1.0
        self.device = "cuda"
        # --- vectorised-engine additions ---
        self.num_envs = 1
        self.updates_per_step = 1
        self.use_graphs = True             # replay the update as one captured hipGraph (train(); update() stays eager)


class Actor(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim=256):
        super().__init__()
        self.fc1 = SmallLinear(state_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc3 = SmallLinear(hidden_dim, action_dim)

    def logits(self, x):
        return self.fc3(self.fc2(self.fc1(x)))

    def forward(self, x):
        return F.softmax(self.logits(x), dim=-1)


class Critic(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim=256):
        super().__init__()
        self.fc1 = SmallLinear(state_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc3 = SmallLinear(hidden_dim, action_dim)

    def forward(self, x):
        return self.fc3(self.fc2(self.fc1(x)))


class SACTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd.sac_cartpole.SACTrainer needs an MI355X and libgymrl_hip.so; no CPU fallback")
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        self.env = VecEnv(config.env_name, config.num_envs, device=self.device, seed=self.base_seed)
        state_dim, action_dim = self.env.observation_space.shape[0], self.env.action_space.n
        self.action_dim = action_dim
        g = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.actor = Actor(state_dim, action_dim, config.hidden_dim)
        self.critic1 = Critic(state_dim, action_dim, config.hidden_dim)
        self.critic2 = Critic(state_dim, action_dim, config.hidden_dim)
        torch.random.set_rng_state(g)
        self.critic1_target, self.critic2_target = copy.deepcopy(self.critic1), copy.deepcopy(self.critic2)
        self.actor_flat, self.actor_grads = flatten_module(self.actor, self.device)
        self.c1_flat, self.c1_grads = flatten_module(self.critic1, self.device)
        self.c2_flat, self.c2_grads = flatten_module(self.critic2, self.device)
        self.c1_target_flat, _ = flatten_module(self.critic1_target, self.device)
        self.c2_target_flat, _ = flatten_module(self.critic2_target, self.device)
        self._actor_sink, self._c1_sink, self._c2_sink = GradSink(self.actor), GradSink(self.critic1), GradSink(self.critic2)
        self.actor_optim = FusedAdam(self.actor_flat, self.actor_grads, lr=config.lr_actor)
        self.critic1_optim = FusedAdam(self.c1_flat, self.c1_grads, lr=config.lr_critic)
        self.critic2_optim = FusedAdam(self.c2_flat, self.c2_grads, lr=config.lr_critic)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.log_alpha = torch.tensor([np.log(0.01)], **f32)                  # float32 scalar (:118-120)
        self._alpha_m, self._alpha_v = torch.zeros(1, **f32), torch.zeros(1, **f32)
        self._alpha_steps = 0
        d64 = dict(dtype=torch.float64, device=self.device)
        self._sums_c, self._sums_a, self._alpha_loss = torch.zeros(2, **d64), torch.zeros(2, **d64), torch.zeros(1, **d64)
        self.memory = ReplayBuffer(config.memory_capacity, state_dim, self.device, seed=self.base_seed)
        self.episode_rewards = deque(maxlen=100)
        self._act_counter = 0
        self._parity_noise = None      # tests: iterator of f32[N, A] Exp(1) draws for select_action
        self._parity_indices = None    # tests: iterator of i32[B] replay indices for update()
        self._graph = None             # hipGraph of the update, captured on first use (update_async)

    @torch.no_grad()
    def select_action(self, state, deterministic=False, noise_exp=None):
        """:127-138 for a batch [N, D] -> i32[N]: argmax of the probabilities or a Categorical draw."""
        state, kind = scalar.obs_batch(state, self.device)       # ONE host observation in -> python int out (sac_cartpole.py:127-138)
        logits = self.actor.logits(state)
        if not deterministic:                        # argmax draws nothing: eval() must not move the exploration stream
            self._act_counter += 1
        act, _, _, _ = ops.categorical_sample(logits, noise_exp=noise_exp, seed=self.base_seed, counter=self._act_counter,
                                              env_id0=self.env.env_id0, deterministic=deterministic)
        return scalar.discrete_out(act, kind)

    def soft_update(self, target_flat, source_flat):
        """:140-145 on the flat parameter buffers."""
        ops.soft_update(target_flat, source_flat, self.cfg.tau)

    def update(self, indices=None):
        """:148-227 -> (actor_loss, critic1_loss, critic2_loss, alpha_loss) python floats."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return 0.0, 0.0, 0.0, 0.0
        if indices is None and self._parity_indices is not None:
            indices = next(self._parity_indices)
        if indices is None:
            indices = self.memory.draw_indices(cfg.batch_size)
        B = self._update_body(indices)
        sc, sa = self._sums_c.tolist(), self._sums_a.tolist()
        return sa[0] / B, sc[0] / B, sc[1] / B, float(self._alpha_loss.item())

    def _update_body(self, indices, biases=None, alpha_bias=None):
        """Everything after the index draw; biases = device views of the three Adams' step scalars (critic1, critic2,
        actor) and alpha_bias the temperature's, when the body runs inside / ahead of a hipGraph."""
        cfg = self.cfg
        bc1, bc2, ba = biases if biases is not None else (None, None, None)
        states, actions, rewards, next_states, dones = self.memory.gather(indices)
        B = states.shape[0]
        with torch.no_grad():                                                  # :171-181
            y = ops.dsac_target(self.actor(next_states), self.critic1_target(next_states),
                                self.critic2_target(next_states), rewards, dones, self.log_alpha, cfg.gamma)
        q1, q2 = self.critic1(states), self.critic2(states)                    # :183-194
        self._sums_c.zero_()
        dq1, dq2 = ops.dsac_critic_loss(q1.detach(), q2.detach(), actions.view(-1).to(torch.int32), y, self._sums_c)
        self._c1_sink.arm()
        self._c2_sink.arm()
        torch.autograd.backward([q1, q2], [dq1, dq2])
        self._c1_sink.collect()
        self._c2_sink.collect()
        self.critic1_optim.step(bias_dev=bc1, polyak=(self.c1_target_flat, cfg.tau))             # + soft updates :217-218
        self.critic2_optim.step(bias_dev=bc2, polyak=(self.c2_target_flat, cfg.tau))
        probs = self.actor(states)                                             # :196-207
        with torch.no_grad():
            q1n, q2n = self.critic1(states), self.critic2(states)              # the critics' gradients of this loss are discarded
        self._sums_a.zero_()
        dprobs = ops.dsac_actor_loss(probs.detach(), q1n, q2n, self.log_alpha, self._sums_a)
        self._actor_sink.arm()
        torch.autograd.backward([probs], [dprobs])
        self._actor_sink.collect()
        self.actor_optim.step(bias_dev=ba)
        if alpha_bias is None:                                                 # :209-215
            self._alpha_steps += 1
        ops.dsac_alpha_step(self.log_alpha, self._alpha_m, self._alpha_v, self._sums_a, B, cfg.target_entropy,
                            cfg.lr_alpha, max(self._alpha_steps, 1), loss_out=self._alpha_loss, bias_dev=alpha_bias)
        return B

    def update_async(self):
        """update() without the host round trip, replayed as a captured hipGraph (gymrl_amd/graphs.py)."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return
        if self._graph is None:
            from .graphs import GraphedUpdate
            b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))          # the kernel's float32 betas, as doubles
            self._graph = GraphedUpdate(self.device, cfg.batch_size, [self.critic1_optim, self.critic2_optim, self.actor_optim],
                                        lambda idx, biases, ab: self._update_body(idx, biases, ab),
                                        alpha=(self, "_alpha_steps", b1, b2))
        self._graph(self.memory, cfg.batch_size)

    def train(self, max_vector_steps=None):
        """The reference's train() loop (every Linear of the update and of acting is a gymrl_lin_* launch: gymrl_amd/nn.py)."""
        return self._train(max_vector_steps)

    def _train(self, max_vector_steps=None):
        """:229-262 with N lock-stepped envs."""
        cfg, env = self.cfg, self.env
        N, D = env.n, env.obs_dim
        obs, nxt, tobs = (torch.empty(N, D, device=self.device) for _ in range(3))
        rew = torch.empty(N, device=self.device)
        tracker = EpisodeTracker(N, self.device, flush_every=1 if N == 1 else 16)
        env.reset(obs)
        step = 0
        graphed = bool(getattr(cfg, "use_graphs", True)) and self._parity_indices is None
        limit = max_vector_steps or (cfg.max_episodes * cfg.max_steps // N + 1)
        while tracker.episodes < cfg.max_episodes and step < limit:
            action = self.select_action(obs, noise_exp=None if self._parity_noise is None else next(self._parity_noise))
            ep_ret, done = tracker.slot()
            env.step(action, nxt, rew, done_out=done, term_obs_out=tobs, ep_ret_out=ep_ret)
            self.memory.push(obs, action, rew, tobs, done)
            if cfg.max_steps < env.max_steps:       # the reference's `for step in range(cfg.max_steps)`: abandoned, no done flag
                env.abandon(cfg.max_steps, nxt, done, ep_ret)
            for _ in range(cfg.updates_per_step):
                if graphed:
                    self.update_async()
                else:
                    self.update()
            obs, nxt = nxt, obs
            step += 1
            tracker.advance(self.episode_rewards)
        tracker.flush(self.episode_rewards)
        self.env.close()

    @torch.no_grad()
    def eval(self, num_episodes=10):
        env = VecEnv(self.cfg.env_name, num_episodes, device=self.device, seed=self.base_seed + 999, env_id0=1 << 40)
        obs = env.reset()
        nxt = torch.empty_like(obs)
        rew = torch.empty(num_episodes, device=self.device)
        done = torch.zeros(num_episodes, dtype=torch.uint8, device=self.device)
        ep_ret = torch.zeros(num_episodes, device=self.device)
        result = torch.full((num_episodes,), float("nan"), device=self.device)
        for _ in range(env.max_steps + 1):
            act = self.select_action(obs, deterministic=True)
            env.step(act, nxt, rew, done_out=done, ep_ret_out=ep_ret)
            result = torch.where(done.bool() & torch.isnan(result), ep_ret, result)
            obs, nxt = nxt, obs
            if not torch.isnan(result).any():
                break
        return result.tolist()

    def test(self):
        return self.eval(num_episodes=5)


if __name__ == "__main__":       # python -m gymrl_amd.sac_cartpole [--<Config attribute> <value> ...]  (sac_cartpole.py:313-329)
    from .utils.cli import run_script
    run_script(Config, SACTrainer)
