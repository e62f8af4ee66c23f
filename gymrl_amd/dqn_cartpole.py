"""DQN — MI355X engine behind the reference's algorithms/dqn_cartpole.py surface:
Config :27-42, QNetwork :53-65, ReplayBuffer :68-88, DQNTrainer :91-253
(get_epsilon :117-122, select_action :124-133, update :135-168, train :170-212).

Underneath: `num_envs` CartPole instances step on the GPU, the replay buffer is a
device-resident SoA ring (append / uniform-index / gather kernels), epsilon-greedy,
the TD target + MSE loss forward/backward and the grad-clamp(+-1) + Adam step are HIP
kernels behind the C-ABI; the three Linear layers run through PyTorch-ROCm.
"""
import copy
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .envs import EpisodeTracker, VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .nn import SmallLinear
from .utils import scalar


class Config:
    def __init__(self):
        self.env_name = "CartPole-v1"
        self.seed = None
        self.max_episodes = 500
        self.max_steps = 500
        self.batch_size = 64
        self.gamma = 0.99
        self.lr = 1e-3
        self.epsilon_start = 0.95
        self.epsilon_end = 0.01
        self.epsilon_decay = 800
        self.target_update_freq = 4          # episodes
        self.memory_capacity = 100000
        self.hidden_dim = 256
        self.device = "cuda"
        # --- vectorised-engine additions (defaults keep the reference's per-step cadence) ---
        self.num_envs = 1
        self.updates_per_step = 1            # reference: one update() per env step (:184)
        self.use_graphs = True               # replay the update as one captured hipGraph (train(); update() stays eager)


def layer_init(layer, std=np.sqrt(2)):
    """dqn_cartpole.py:44-49: orthogonal weights (gain std), zero bias."""
    if isinstance(layer, nn.Linear):
        nn.init.orthogonal_(layer.weight, gain=std)
        if layer.bias is not None:
            nn.init.constant_(layer.bias, 0)
    return layer


class QNetwork(nn.Module):
    """dqn_cartpole.py:53-65 (same module tree, so reference state_dicts load unchanged)."""

    def __init__(self, state_dim, action_dim, hidden_dim=256):
        super().__init__()
        self.net = nn.Sequential(
            layer_init(SmallLinear(state_dim, hidden_dim, act="relu")), nn.Identity(),      # the ReLUs run inside the layers'
            layer_init(SmallLinear(hidden_dim, hidden_dim, act="relu")), nn.Identity(),     # launches (csrc/lin.hip)
            layer_init(SmallLinear(hidden_dim, action_dim), std=0.01),
        )

    def forward(self, x):
        return self.net(x)


class ReplayBuffer:
    """Device ring replacing deque(maxlen) + random.sample (:68-88).  push() takes N rows."""

    def __init__(self, capacity, state_dim, device, action_words=1, action_dtype=torch.int32, seed=0):
        self.capacity, self.device, self.action_dtype = int(capacity), device, action_dtype
        self.ring = (torch.zeros(capacity, state_dim, device=device),
                     torch.zeros(capacity, action_words, dtype=torch.int32, device=device),
                     torch.zeros(capacity, device=device), torch.zeros(capacity, state_dim, device=device),
                     torch.zeros(capacity, dtype=torch.uint8, device=device))
        self.cursor, self.size, self.seed, self.draws = 0, 0, seed, 0

    def push(self, state, action, reward, next_state, done, cursor_dev=None):
        """cursor_dev (StepChunk capture): device int64[1] cursor of the replayed step; the host cursor is advanced by
        the trainer's staging loop (advance()) instead."""
        if not (torch.is_tensor(reward) and reward.is_cuda):
            # host scalars / numpy rows — the reference's `memory.push(state, action, reward, next_state, done)` (:183)
            n = int(np.size(reward.cpu().numpy() if torch.is_tensor(reward) else reward))
            D, AW = self.ring[0].shape[1], self.ring[1].shape[1]
            state, next_state = (scalar.rows(x, n, torch.float32, self.device, D) for x in (state, next_state))
            action = scalar.rows(action, n, self.action_dtype, self.device, AW)
            reward, done = scalar.rows(reward, n, torch.float32, self.device), scalar.rows(done, n, torch.uint8, self.device)
        n = reward.numel()
        if action.dtype == torch.float32 and self.action_dtype == torch.float32:
            action = action.contiguous().view(torch.int32)    # float32 action words (SAC / TD3 / DDPG rings), reinterpreted
        elif action.dtype != torch.int32:
            if self.action_dtype == torch.float32 or action.is_floating_point():
                raise TypeError(f"replay ring of {self.action_dtype} actions got a {action.dtype} action tensor")
            action = action.to(torch.int32)                   # discrete ring: an int64 tensor (the reference's dtype) is converted
        ops.replay_append(self.ring, self.cursor, state, action.view(n, -1), reward, next_state, done, cursor_dev=cursor_dev)
        if cursor_dev is None:
            self.advance(n)

    def advance(self, n):
        self.cursor = (self.cursor + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def draw_indices(self, batch_size, out=None, dev=None):
        """random.sample's role: one uniform index draw per call (counter-keyed Philox); `out` = fixed buffer.
        dev (StepChunk capture): device record {counter, size} of the replayed step."""
        if dev is not None:
            return ops.uniform_indices(self.seed, 0, self.capacity, batch_size, self.device, out=out, dev=dev)
        idx = ops.uniform_indices(self.seed, self.draws, self.size, min(batch_size, self.size), self.device, out=out)
        self.draws += 1
        return idx

    def gather(self, indices):
        return ops.replay_gather(self.ring, indices, self.action_dtype)

    def sample(self, batch_size, indices=None):
        """-> (states, actions, rewards, next_states, dones f32).  `indices` replays an explicit draw."""
        return self.gather(self.draw_indices(batch_size) if indices is None else indices)

    def __len__(self):
        return self.size

    def state_dict(self):
        """Ring contents + cursors, so that a resumed run samples what the interrupted one would have."""
        return {"ring": [t.detach().cpu() for t in self.ring], "cursor": self.cursor, "size": self.size, "draws": self.draws}

    def load_state_dict(self, sd):
        for dst, src in zip(self.ring, sd["ring"]):
            dst.copy_(src.to(dst.device))
        self.cursor, self.size, self.draws = int(sd["cursor"]), int(sd["size"]), int(sd["draws"])


class DQNTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd.DQNTrainer needs an MI355X and libgymrl_hip.so; no CPU fallback")
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        self.env = VecEnv(config.env_name, config.num_envs, device=self.device, seed=self.base_seed)
        state_dim, action_dim = self.env.observation_space.shape[0], self.env.action_space.n
        self.action_dim = action_dim
        g = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.policy_net = QNetwork(state_dim, action_dim, config.hidden_dim)
        torch.random.set_rng_state(g)
        self.target_net = copy.deepcopy(self.policy_net)
        self.flat_params, self.flat_grads = flatten_module(self.policy_net, self.device)
        self.target_flat, _ = flatten_module(self.target_net, self.device)
        self.target_net.eval()
        self._sink = GradSink(self.policy_net)
        self.optimizer = FusedAdam(self.flat_params, self.flat_grads, lr=config.lr, eps=1e-8, clamp_abs=1.0)
        self.memory = ReplayBuffer(config.memory_capacity, state_dim, self.device, seed=self.base_seed)
        self.epsilon = config.epsilon_start
        self.sample_count = 0
        self.episode_rewards = deque(maxlen=100)
        self._loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._act_counter = 0
        self._parity_u = None          # tests: iterator of f32[N, 2] uniforms for select_action (explore?, which action)
        self._parity_indices = None    # tests: iterator of i32[B] replay indices for update()
        self._graph = None             # hipGraph of the update, captured on first use (update_async)

    def get_epsilon(self):
        """:117-122 — advanced once per (vector) action selection."""
        self.sample_count += 1
        self.epsilon = self.cfg.epsilon_end + (self.cfg.epsilon_start - self.cfg.epsilon_end) * np.exp(
            -1.0 * self.sample_count / self.cfg.epsilon_decay)
        return self.epsilon

    @torch.no_grad()
    def select_action(self, state, deterministic=False, u=None):
        """:124-133 for a batch of states [N, D] -> i32[N] (device in, device out).  The reference's scalar surface: ONE host
        observation (np.ndarray [D]) in -> python int out, as `select_action(state) -> int` at :124-133."""
        state, kind = scalar.obs_batch(state, self.device)
        q = self.policy_net(state)
        if deterministic:       # eval(): greedy, and the exploration stream must not move (bit-exact resume / replay)
            return scalar.discrete_out(ops.epsilon_greedy(q, 0.0, u=u, seed=self.base_seed, counter=0, env_id0=self.env.env_id0), kind)
        eps = self.get_epsilon()
        self._act_counter += 1
        return scalar.discrete_out(ops.epsilon_greedy(q, eps, u=u, seed=self.base_seed, counter=self._act_counter,
                                                      env_id0=self.env.env_id0), kind)

    def load_target(self):
        self.target_flat.copy_(self.flat_params)        # target_net.load_state_dict(policy_net.state_dict())

    def update(self, indices=None):
        """:135-168.  Returns the loss as a python float (one host sync, like loss.item())."""
        if len(self.memory) < self.cfg.batch_size:
            return 0.0
        if indices is None and self._parity_indices is not None:
            indices = next(self._parity_indices)
        if indices is None:
            indices = self.memory.draw_indices(self.cfg.batch_size)
        B = self._update_body(indices)
        return float(self._loss.item()) / B

    def _update_body(self, indices, bias=None):
        """Everything after the index draw; bias = f32[4] device view of Adam's step scalars under a hipGraph."""
        states, actions, rewards, next_states, dones = self.memory.gather(indices)
        q = self.policy_net(states)
        with torch.no_grad():
            qn = self.target_net(next_states)
        self._loss.zero_()
        td, dq = ops.dqn_td_loss(q, qn, actions.view(-1), rewards, dones, self.cfg.gamma, loss_sum=self._loss)
        self._sink.arm()
        q.backward(dq)
        self._sink.collect()
        self.optimizer.step(bias_dev=bias)               # grad clamp +-1 (:163-165) fused into the Adam kernel
        return states.shape[0]

    def update_async(self):
        """update() without the host round trip: eager index draw + scalar store, then the captured hipGraph of
        `_update_body` (gymrl_amd/graphs.py).  The loss sum stays on the device (`_loss`)."""
        cfg, m = self.cfg, self.memory
        if len(m) < cfg.batch_size:
            return
        if self._graph is None:
            from .graphs import GraphedStep, StepScalars
            self._scalars = StepScalars(self.device)
            bias, self._off = self._scalars.slot(16, torch.float32)
            self._g_idx = torch.empty(cfg.batch_size, dtype=torch.int32, device=self.device)
            self._graph = GraphedStep(lambda: self._update_body(self._g_idx, bias=bias))
        m.draw_indices(cfg.batch_size, out=self._g_idx)
        self._scalars.set(self._off, self.optimizer.next_bias())
        self._scalars.flush()
        self._graph()

    def save_checkpoint(self, path, include_memory=True):
        """ModelLoader-style dict: networks, the optimiser in torch.optim.Adam's layout, the
        schedule counters and — unlike the reference, which skips `memory` — the replay ring."""
        from .utils import checkpoint
        extra = {"memory_state_dict": self.memory.state_dict()} if include_memory else {}
        return checkpoint.save_agent(path, {"policy_net": self.policy_net, "target_net": self.target_net},
                                     {"optimizer": (self.policy_net, self.optimizer)}, epsilon=self.epsilon,
                                     sample_count=self.sample_count, _act_counter=self._act_counter,
                                     episode_rewards=list(self.episode_rewards), **extra)

    def load_checkpoint(self, path):
        from .utils import checkpoint
        rest = checkpoint.load_agent(path, {"policy_net": self.policy_net, "target_net": self.target_net},
                                     {"optimizer": (self.policy_net, self.optimizer)})
        self.epsilon, self.sample_count = float(rest["epsilon"]), int(rest["sample_count"])
        self._act_counter = int(rest["_act_counter"])
        self.episode_rewards.clear()
        self.episode_rewards.extend(rest.get("episode_rewards", []))
        if "memory_state_dict" in rest:
            self.memory.load_state_dict(rest["memory_state_dict"])
        return rest

    def train(self, max_vector_steps=None):
        """The reference's train() loop (every Linear of the update and of acting is a gymrl_lin_* launch: gymrl_amd/nn.py)."""
        return self._train(max_vector_steps)

    def _train(self, max_vector_steps=None):
        """:170-212 with N lock-stepped envs; "episodes" counts finished episodes over all envs."""
        cfg, env = self.cfg, self.env
        N, D = env.n, env.obs_dim
        obs, nxt, tobs = (torch.empty(N, D, device=self.device) for _ in range(3))
        rew = torch.empty(N, device=self.device)
        tracker = EpisodeTracker(N, self.device, flush_every=1 if N == 1 else 16)
        env.reset(obs)
        step, last_target = 0, 0
        graphed = bool(getattr(cfg, "use_graphs", True)) and self._parity_indices is None
        limit = max_vector_steps or (cfg.max_episodes * cfg.max_steps // N + 1)
        while tracker.episodes < cfg.max_episodes and step < limit:
            action = self.select_action(obs, u=None if self._parity_u is None else next(self._parity_u))
            ep_ret, done = tracker.slot()
            env.step(action, nxt, rew, done_out=done, term_obs_out=tobs, ep_ret_out=ep_ret)
            self.memory.push(obs, action, rew, tobs, done)      # next_state = pre-reset observation (:183)
            if cfg.max_steps < env.max_steps:       # :178 `for step in range(cfg.max_steps)`: the episode is abandoned without a
                env.abandon(cfg.max_steps, nxt, done, ep_ret)     # done flag (stored above) and the next one starts
            for _ in range(cfg.updates_per_step):
                if graphed:
                    self.update_async()
                else:
                    self.update()
            obs, nxt = nxt, obs
            step += 1
            tracker.advance(self.episode_rewards)
            if tracker.episodes - last_target >= cfg.target_update_freq:    # :193-194
                self.load_target()
                last_target = tracker.episodes
            if len(self.episode_rewards) >= 100 and np.mean(self.episode_rewards) >= 495.0:
                break
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


if __name__ == "__main__":       # python -m gymrl_amd.dqn_cartpole [--<Config attribute> <value> ...]  (dqn_cartpole.py:256-272)
    from .utils.cli import run_script
    run_script(Config, DQNTrainer)
