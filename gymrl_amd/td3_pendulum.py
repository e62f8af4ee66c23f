"""TD3 (twin critics, delayed policy updates, target-policy smoothing) — MI355X engine behind the
reference's algorithms/td3_pendulum.py surface: Config :28-45, Actor :48-61, Critic :64-92 (both Q
networks in one module + q1()), ReplayBuffer :95-116, TD3Trainer :120-329 (soft_update :149-154,
select_action :156-169, update :171-228, train / eval / test).

Underneath: Pendulum instances step on the GPU; replay ring, exploration and smoothing noise
(`gymrl_noisy_action`), the Bellman target (`gymrl_sac_target` with zero log-prob), the twin-critic
loss (`gymrl_sac_critic_loss`), the actor loss (`gymrl_neg_mean_loss`), the fused Adam steps and
the Polyak updates are HIP kernels behind the C-ABI; Linear layers run through PyTorch-ROCm.
"""
import copy
from collections import deque

import torch
import torch.nn as nn

from . import ops
from .envs import EpisodeTracker, VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .nn import SmallLinear, frozen_parameters, fused_linears
from .sac_pendulum import ReplayBuffer
from .utils import scalar


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
        self.policy_noise = 0.2
        self.noise_clip = 0.5
        self.exploration_noise = 0.1
        self.policy_freq = 2
        self.memory_capacity = 100000
        self.hidden_dim = 256
        self.device = "cuda"
        # --- vectorised-engine additions ---
        self.num_envs = 1
        self.updates_per_step = 1


class Actor(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim, action_bound):
        super().__init__()
        self.action_bound = action_bound
        self.fc1 = SmallLinear(state_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc3 = SmallLinear(hidden_dim, action_dim, act="tanh")

    def forward(self, x):
        return self.fc3(self.fc2(self.fc1(x))) * self.action_bound


class Critic(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim):
        super().__init__()
        self.fc1 = SmallLinear(state_dim + action_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc3 = SmallLinear(hidden_dim, 1)
        self.fc4 = SmallLinear(state_dim + action_dim, hidden_dim, act="relu")
        self.fc5 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc6 = SmallLinear(hidden_dim, 1)

    def forward(self, state, action):
        """Both Q networks, the twins sharing each layer's launch; cat([state, action]) is never materialised."""
        h1, h4 = fused_linears([self.fc1, self.fc4], [state, state], [action, action])
        h2, h5 = fused_linears([self.fc2, self.fc5], [h1, h4])
        q1, q2 = fused_linears([self.fc3, self.fc6], [h2, h5])
        return q1, q2

    def q1(self, state, action):
        return self.fc3(self.fc2(self.fc1(state, action)))


class _ActorCriticBase:
    """What TD3 and DDPG share: env, replay ring, exploration, vectorised train / eval loops."""

    def _setup(self, config, critic_cls):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError(f"gymrl_amd.{type(self).__name__} needs an MI355X and libgymrl_hip.so; no CPU fallback")
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        self.env = VecEnv(config.env_name, config.num_envs, device=self.device, seed=self.base_seed)
        state_dim, action_dim = self.env.observation_space.shape[0], self.env.action_space.shape[0]
        self.action_bound = float(self.env.action_space.high[0])
        g = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.actor = Actor(state_dim, action_dim, config.hidden_dim, self.action_bound)
        self.critic = critic_cls(state_dim, action_dim, config.hidden_dim)
        torch.random.set_rng_state(g)
        self.actor_target = copy.deepcopy(self.actor)
        self.critic_target = copy.deepcopy(self.critic)
        self.actor_flat, self.actor_grads = flatten_module(self.actor, self.device)
        self.critic_flat, self.critic_grads = flatten_module(self.critic, self.device)
        self.actor_target_flat, _ = flatten_module(self.actor_target, self.device)
        self.critic_target_flat, _ = flatten_module(self.critic_target, self.device)
        self._actor_sink, self._critic_sink = GradSink(self.actor), GradSink(self.critic)
        self.actor_optimizer = FusedAdam(self.actor_flat, self.actor_grads, lr=config.lr_actor)
        self.critic_optimizer = FusedAdam(self.critic_flat, self.critic_grads, lr=config.lr_critic)
        self.memory = ReplayBuffer(config.memory_capacity, state_dim, action_dim, self.device, seed=self.base_seed)
        self.episode_rewards = deque(maxlen=100)
        d64 = dict(dtype=torch.float64, device=self.device)
        self._sum_c, self._sum_a = torch.zeros(1, **d64), torch.zeros(1, **d64)
        self._log_alpha0 = torch.zeros(1, **d64)     # alpha * 0 log-prob: the SAC target kernel as a plain TD target
        self._act_counter = 0
        self._noise_counter = 0
        self._parity_eps = None        # tests: iterator of f64[N, A] N(0,1) draws for select_action
        self._parity_updates = None    # tests: iterator of per-update tuples (see update())
        self._graph = None             # hipGraph of the update (trainers that define update_async)

    def soft_update(self, target_flat, source_flat):
        """:149-154 on the flat parameter buffers."""
        ops.soft_update(target_flat, source_flat, self.cfg.tau)

    @torch.no_grad()
    def select_action(self, state, deterministic=False, eps=None):
        """select_action for a batch [N, D] -> f32[N, A]: actor + clipped Gaussian exploration noise."""
        state, kind = scalar.obs_batch(state, self.device)       # ONE host observation in -> np.ndarray [act_dim] out
        action = self.actor(state)
        if deterministic:
            return scalar.continuous_out(action, kind)
        self._act_counter += 1
        return scalar.continuous_out(ops.noisy_action(action.contiguous(), self._exploration_std() * self.action_bound,
                                                      self.action_bound, eps=eps, mode=0, seed=self.base_seed,
                                                      counter=self._act_counter), kind)

    def train(self, max_vector_steps=None):
        """The reference's train() loop (every Linear of the update and of acting is a gymrl_lin_* launch: gymrl_amd/nn.py)."""
        return self._train(max_vector_steps)

    def _train(self, max_vector_steps=None):
        """The reference's episode loop with N lock-stepped envs: act, step, push, update every step."""
        cfg, env = self.cfg, self.env
        N, D = env.n, env.obs_dim
        obs, nxt, tobs = (torch.empty(N, D, device=self.device) for _ in range(3))
        rew = torch.empty(N, device=self.device)
        tracker = EpisodeTracker(N, self.device, flush_every=1 if N == 1 else 16)
        env.reset(obs)
        step = 0
        # DDPG replays its update as a captured hipGraph; TD3's update draws its target-smoothing noise from a host
        # counter and alternates between two bodies (policy delay), and stays eager
        graphed = (bool(getattr(cfg, "use_graphs", True)) and self._parity_updates is None and hasattr(self, "update_async"))
        limit = max_vector_steps or (cfg.max_episodes * cfg.max_steps // N + 1)
        while tracker.episodes < cfg.max_episodes and step < limit:
            action = self.select_action(obs, eps=None if self._parity_eps is None else next(self._parity_eps))
            ep_ret, done = tracker.slot()
            env.step(action, nxt, rew, done_out=done, term_obs_out=tobs, ep_ret_out=ep_ret)
            self.memory.push(obs, action, rew, tobs, done)             # done = terminated or truncated
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
            env.step(act.contiguous(), nxt, rew, done_out=done, ep_ret_out=ep_ret)
            result = torch.where(done.bool() & torch.isnan(result), ep_ret, result)
            obs, nxt = nxt, obs
            if not torch.isnan(result).any():
                break
        return result.tolist()

    def test(self):
        return self.eval(num_episodes=5)


class TD3Trainer(_ActorCriticBase):
    def __init__(self, config):
        self._setup(config, Critic)
        self.total_updates = 0

    def _exploration_std(self):
        return self.cfg.exploration_noise

    def update(self, indices=None, eps=None):
        """:171-228 -> (actor_loss, critic_loss) python floats (actor_loss 0.0 on the skipped policy steps)."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return 0.0, 0.0
        self.total_updates += 1
        if indices is None and self._parity_updates is not None:
            indices, eps = next(self._parity_updates)
        states, actions, rewards, next_states, dones = self.memory.sample(cfg.batch_size, indices)
        B = states.shape[0]
        with torch.no_grad():                                          # :191-199
            self._noise_counter += 1
            next_actions = ops.noisy_action(self.actor_target(next_states).contiguous(), cfg.policy_noise,
                                            self.action_bound, eps=eps, mode=1, noise_clip=cfg.noise_clip,
                                            seed=self.base_seed + 1, counter=self._noise_counter)
            tq1, tq2 = self.critic_target(next_states, next_actions)
            y = ops.sac_target(rewards, dones, tq1.view(-1), tq2.view(-1), torch.zeros_like(rewards),
                               self._log_alpha0, cfg.gamma)
        q1, q2 = self.critic(states, actions)                          # :201-208
        self._sum_c.zero_()
        dq1, dq2 = ops.sac_critic_loss(q1.view(-1), q2.view(-1), y, self._sum_c)
        self._critic_sink.arm()                                        # critic_optimizer.zero_grad()
        torch.autograd.backward([q1, q2], [dq1.view_as(q1), dq2.view_as(q2)])
        self._critic_sink.collect()
        delayed = self.total_updates % cfg.policy_freq == 0
        self.critic_optimizer.step(polyak=(self.critic_target_flat, cfg.tau) if delayed else None)   # + soft update :223 when due
        actor_loss = 0.0
        if delayed:                                                    # :210-224
            with frozen_parameters(self.critic):                       # its gradients of this loss are never used (:218 zero_grad)
                q = self.critic.q1(states, self.actor(states))
            self._sum_a.zero_()
            dq = ops.neg_mean_loss(q.view(-1), self._sum_a)
            self._actor_sink.arm()
            torch.autograd.backward([q], [dq.view_as(q)])
            self._actor_sink.collect()
            self.actor_optimizer.step(polyak=(self.actor_target_flat, cfg.tau))
            actor_loss = -float(self._sum_a.item()) / B
        return actor_loss, float(self._sum_c.item()) / B


if __name__ == "__main__":       # python -m gymrl_amd.td3_pendulum [--<Config attribute> <value> ...]  (td3_pendulum.py:313-329)
    from .utils.cli import run_script
    run_script(Config, TD3Trainer)
