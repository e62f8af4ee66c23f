"""SAC (twin-Q, tanh-Gaussian reparameterised actor, automatic temperature) — MI355X engine
behind the reference's algorithms/sac_pendulum.py surface: Config :29-46, Actor :49-98,
Critic :101-125, ReplayBuffer :128-148, SACTrainer :151-351 (soft_update :194-199,
select_action :202-211, update :213-267, train :269-310).

Underneath: Pendulum instances step on the GPU; replay ring, the reparameterised sample
+ log-prob (forward AND backward), the soft-Bellman target, the critic / actor losses'
forward+backward, the float64 log_alpha Adam step, the three fused Adam steps and the
Polyak update are HIP kernels behind the C-ABI; Linear layers run through PyTorch-ROCm.
"""
import copy
import os
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .dqn_cartpole import ReplayBuffer as _Ring
from .envs import EpisodeTracker, VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .nn import SmallLinear, frozen_parameters, fused_linears
from .utils import scalar


class Config:
    def __init__(self):
        self.env_name = "Pendulum-v1"
        self.seed = None
        self.max_episodes = 500
        self.max_steps = 200
        self.batch_size = 128
        self.gamma = 0.99
        self.lr_actor = 3e-4
        self.lr_critic = 3e-4
        self.lr_alpha = 3e-4
        self.tau = 0.005
        self.init_alpha = 0.2
        self.memory_capacity = 100000
        self.hidden_dim = 256
        self.log_std_min = -20
        self.log_std_max = 2
        self.device = "cuda"
        # --- vectorised-engine additions ---
        self.num_envs = 1
        self.updates_per_step = 1
        self.use_graphs = True             # replay the update as one captured hipGraph (train(); update() stays eager)
        self.fused_images = True           # fused step: keep MFMA-operand images of the hidden x hidden weights (3x faster to stream)
        self.fused_step = True             # a vector step = 5 launches (csrc/offpolicy_step.hip: acting + env + append in one, the
        #                                    update in four) instead of ~60; bit-identical to the layer-by-layer path, which remains
        #                                    for shapes beyond the kernels' limits (hidden_dim > 256, batch > 256) and custom envs
        self.one_launch_step = os.environ.get("GYMRL_SAC_ONE_LAUNCH", "0") != "0"   # the vector step as ONE launch (gymrl_sac_step: the five
        #                                    phases as block ranges of one grid, handing over through counters in the workspace;
        #                                    bit-identical).  Measured slower than the five launches (0.123 vs 0.100 ms): off
        self.chunk_steps = 0               # 16: whole vector steps as one hipGraph per 16 (graphs.StepChunk).  Measured slower
        #                                    here (0.37 vs 0.29 ms per step at N = 4096, B = 128): SAC's step is GPU-bound, and
        #                                    the executor's cost per graph node grows with the graph (944 nodes per chunk)


class _SacSample(torch.autograd.Function):
    """Actor.sample's tail (:78-87) as one HIP kernel each way."""

    @staticmethod
    def forward(ctx, mean, log_std, eps, bound):
        mean, log_std = mean.contiguous(), log_std.contiguous()
        action, logp = ops.sac_sample_fwd(mean, log_std, eps, bound)
        ctx.save_for_backward(mean, log_std, eps)
        ctx.bound = bound
        return action, logp.unsqueeze(1)

    @staticmethod
    def backward(ctx, d_action, d_logp):
        mean, log_std, eps = ctx.saved_tensors
        da = None if d_action is None else d_action.contiguous()
        dl = None if d_logp is None else d_logp.reshape(-1).contiguous()
        dm, ds = ops.sac_sample_bwd(mean, log_std, eps, da, dl, ctx.bound)
        return dm, ds, None, None


class Actor(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim, action_bound, log_std_min, log_std_max):
        super().__init__()
        self.action_bound, self.log_std_min, self.log_std_max = action_bound, log_std_min, log_std_max
        self.fc1 = SmallLinear(state_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.mean = SmallLinear(hidden_dim, action_dim)
        self.log_std = SmallLinear(hidden_dim, action_dim, act="clamp", clamp=(log_std_min, log_std_max))

    def forward(self, x):
        x = self.fc2(self.fc1(x))                                   # Linear + ReLU per launch (csrc/lin.hip)
        mean, log_std = fused_linears([self.mean, self.log_std], [x, x])      # both heads (log_std clamped) in one launch
        return mean, log_std

    def sample(self, state, eps=None):
        """:76-87 -> (action [B, A], log_prob [B, 1]).  eps: explicit N(0,1) draws (parity mode)."""
        mean, log_std = self.forward(state)
        if eps is None:
            eps = torch.randn_like(mean)
        return _SacSample.apply(mean, log_std, eps, float(self.action_bound))

    @torch.no_grad()
    def get_action(self, state, deterministic=False, eps=None):
        mean, log_std = self.forward(state)
        if deterministic:
            return torch.tanh(mean) * self.action_bound
        if eps is None:
            eps = torch.randn_like(mean)
        return ops.sac_sample_fwd(mean.contiguous(), log_std.contiguous(), eps, float(self.action_bound))[0]


class Critic(nn.Module):
    """Both Q networks in one module (:101-125)."""

    def __init__(self, state_dim, action_dim, hidden_dim):
        super().__init__()
        self.fc1 = SmallLinear(state_dim + action_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc3 = SmallLinear(hidden_dim, 1)
        self.fc4 = SmallLinear(state_dim + action_dim, hidden_dim, act="relu")
        self.fc5 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.fc6 = SmallLinear(hidden_dim, 1)

    def forward(self, state, action):
        """Both Q networks layer by layer, the twins sharing each launch; cat([state, action]) is never materialised."""
        h1, h4 = fused_linears([self.fc1, self.fc4], [state, state], [action, action])
        h2, h5 = fused_linears([self.fc2, self.fc5], [h1, h4])
        q1, q2 = fused_linears([self.fc3, self.fc6], [h2, h5])
        return q1, q2


class ReplayBuffer(_Ring):
    """Device ring with float32 action words (:128-148)."""

    def __init__(self, capacity, state_dim, action_dim, device, seed=0):
        super().__init__(capacity, state_dim, device, action_words=action_dim, action_dtype=torch.float32, seed=seed)

    def push(self, state, action, reward, next_state, done, cursor_dev=None):
        super().push(state, action, reward, next_state, done, cursor_dev=cursor_dev)     # the ring stores the f32 bits as words


class SACTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd.SACTrainer needs an MI355X and libgymrl_hip.so; no CPU fallback")
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        self.env = VecEnv(config.env_name, config.num_envs, device=self.device, seed=self.base_seed)
        state_dim, action_dim = self.env.observation_space.shape[0], self.env.action_space.shape[0]
        self.action_bound = float(self.env.action_space.high[0])
        g = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.actor = Actor(state_dim, action_dim, config.hidden_dim, self.action_bound, config.log_std_min,
                           config.log_std_max)
        self.critic = Critic(state_dim, action_dim, config.hidden_dim)
        torch.random.set_rng_state(g)
        self.critic_target = copy.deepcopy(self.critic)
        self.actor_flat, self.actor_grads = flatten_module(self.actor, self.device)
        self.critic_flat, self.critic_grads = flatten_module(self.critic, self.device)
        self.critic_target_flat, _ = flatten_module(self.critic_target, self.device)
        self._actor_sink, self._critic_sink = GradSink(self.actor), GradSink(self.critic)
        self.actor_optimizer = FusedAdam(self.actor_flat, self.actor_grads, lr=config.lr_actor)
        self.critic_optimizer = FusedAdam(self.critic_flat, self.critic_grads, lr=config.lr_critic)
        self.target_entropy = -action_dim
        d64 = dict(dtype=torch.float64, device=self.device)
        self.log_alpha = torch.tensor([np.log(config.init_alpha)], **d64)      # float64 like the reference (:177-179)
        self._alpha_m, self._alpha_v = torch.zeros(1, **d64), torch.zeros(1, **d64)
        self._alpha_steps = 0
        self._sums = torch.zeros(4, **d64)
        self._alpha_loss = torch.zeros(1, **d64)
        self.memory = ReplayBuffer(config.memory_capacity, state_dim, action_dim, self.device, seed=self.base_seed)
        self.episode_rewards = deque(maxlen=100)
        self._parity_eps = None        # tests: iterator of f32[N, A] N(0,1) draws for select_action
        self._graph = None             # hipGraph of the update, captured on first use (update_async)
        self._parity_updates = None    # tests: iterator of (indices i32[B], eps_next [B, A], eps_cur [B, A]) for update()
        self._fused = None             # (act args, update args, workspace, env, weight images) of the fused step, built on first use
        self._img_versions = None      # versions of the flat buffers the weight images were last rebuilt from
        self._act_noise = self._upd_noise = 0      # Philox counters of the fused step's own N(0,1) draws

    @property
    def alpha(self):
        return self.log_alpha.exp()

    def soft_update(self, target_flat=None, source_flat=None):
        """:194-199 on the flat parameter buffers."""
        ops.soft_update(self.critic_target_flat if target_flat is None else target_flat,
                        self.critic_flat if source_flat is None else source_flat, self.cfg.tau)
        self._img_versions = None             # a raw-pointer write: the fused step's weight images of the target are stale

    # ------------------------------------------------------------ fused vector step (csrc/offpolicy_step.hip) --
    def _fused_update_ok(self):
        """update() as gymrl_sac_update: a matter of shapes only (any env, any source of transitions)."""
        cfg, m = self.cfg, self.memory
        return (bool(getattr(cfg, "fused_step", True))
                and ops.sac_fused_shape_ok(cfg.batch_size, m.ring[0].shape[1], m.ring[1].shape[1], cfg.hidden_dim))

    def _fused_ok(self):
        """The whole vector step fused: the update AND acting + env step + replay row (gymrl_sac_act_step steps Pendulum itself)."""
        cfg, env = self.cfg, self.env
        return (self._fused_update_ok() and isinstance(env, VecEnv) and env.kind == ops.PENDULUM
                and cfg.max_steps >= env.max_steps and self.memory.capacity >= env.n)

    def _fused_args(self):
        if self._fused is None or self._fused[3] is not self.env:
            cfg, env, m = self.cfg, self.env, self.memory
            D, A = m.ring[0].shape[1], m.ring[1].shape[1]
            img = ops.sac_images(cfg.hidden_dim, self.device) if getattr(cfg, "fused_images", True) else None
            act = (ops.sac_act_args(env, self.actor, m.ring, m.capacity, self.action_bound, cfg.log_std_min, cfg.log_std_max, img)
                   if isinstance(env, VecEnv) else None)
            ws = ops.sac_update_workspace(cfg.batch_size, D, A, cfg.hidden_dim, self.device)
            upd = ops.sac_update_args(cfg.batch_size, D, A, self.actor, self.critic, self.critic_target,
                                      self.actor_optimizer, self.critic_optimizer, m.ring,
                                      (cfg.gamma, cfg.tau, self.action_bound, cfg.log_std_min, cfg.log_std_max, self.target_entropy,
                                       cfg.lr_alpha), self.log_alpha, self._alpha_m, self._alpha_v, self._sums, self._alpha_loss, ws, img)
            self._fused = (act, upd, ws, env, img)
            self._img_versions = None
        # the weight images follow the parameters as long as only the fused update writes them; anything that went through
        # torch (load_state_dict, a checkpoint, a hard target copy: the flat buffers' version counters move) or through the
        # layer-by-layer update (which resets _img_versions) makes them stale: rebuild (one launch)
        if self._fused[4] is not None:
            # (flatten_module binds every parameter as a VIEW tensor of its own: load_state_dict / a checkpoint bump the
            # parameters' version counters, not the flat buffers' — so both are summed; writers that go around torch
            # altogether, soft_update() and load_checkpoint(), reset _img_versions themselves)
            ps = getattr(self, "_img_params", None)
            if ps is None:
                ps = self._img_params = tuple([f] + list(net.parameters()) for f, net in (
                    (self.actor_flat, self.actor), (self.critic_flat, self.critic), (self.critic_target_flat, self.critic_target)))
            v = tuple(sum([t._version for t in group]) for group in ps)
            if v != self._img_versions:
                ops.sac_pack_images(self._fused[1])
                self._img_versions = v
        return self._fused

    def _update_fused(self, indices=None, eps_next=None, eps_cur=None, dev=None, launch=True):
        """update() as gymrl_sac_update's four launches.  dev = (draw, adam_c, adam_a, alpha, noise) device records of a
        StepChunk replay; None: this call's scalars travel as arguments and the host counters advance here.  launch=False: the
        arguments are prepared only (ops.sac_step issues them with the acting step's)."""
        m = self.memory
        upd = self._fused_args()[1]
        if dev is not None:
            ops.sac_update(upd, idx_seed=m.seed, idx_dev=dev[0], adam_critic_dev=dev[1], adam_actor_dev=dev[2], alpha_bias_dev=dev[3],
                           noise_seed=self.base_seed, noise_counter_dev=dev[4], idx_size=m.capacity, launch=launch)
            return
        if indices is None:
            counter, size = m.draws, m.size
            m.draws += 1
        else:
            counter, size = 0, 0
        self._alpha_steps += 1
        self._upd_noise += 1
        t = self._alpha_steps
        ops.sac_update(upd, idx=indices, idx_seed=m.seed, idx_counter=counter, idx_size=size, eps_next=eps_next, eps_cur=eps_cur,
                       noise_seed=self.base_seed, noise_counter=self._upd_noise, adam_critic=self.critic_optimizer.next_bias(),
                       adam_actor=self.actor_optimizer.next_bias(), alpha_bias=(1.0 - 0.9 ** t, 1.0 - 0.999 ** t), launch=launch)

    @torch.no_grad()
    def select_action(self, state, deterministic=False, eps=None):
        """:202-211 for a batch [N, D] -> f32[N, A] (stays on the device).  The reference's scalar surface: ONE host
        observation (np.ndarray [D]) in -> np.ndarray [act_dim] out (`.cpu().numpy().flatten()`, :211)."""
        state, kind = scalar.obs_batch(state, self.device)
        return scalar.continuous_out(self.actor.get_action(state, deterministic, eps), kind)

    def update(self, indices=None, eps_next=None, eps_cur=None):
        """:213-267 -> (actor_loss, critic_loss, alpha_loss) python floats."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return 0.0, 0.0, 0.0
        if indices is None and self._parity_updates is not None:
            indices, eps_next, eps_cur = next(self._parity_updates)
        if self._fused_update_ok() and (indices is None or indices.numel() == cfg.batch_size):
            self._update_fused(indices, eps_next, eps_cur)
            s = self._sums.tolist()
            return s[1] / cfg.batch_size, s[0] / cfg.batch_size, float(self._alpha_loss.item())
        if indices is None:
            indices = self.memory.draw_indices(cfg.batch_size)
        B = self._update_body(indices, eps_next, eps_cur)
        s = self._sums.tolist()
        return s[1] / B, s[0] / B, float(self._alpha_loss.item())

    def _update_body(self, indices, eps_next=None, eps_cur=None, bias=None):
        """Everything after the index draw.  bias = (critic f32[4], actor f32[4], alpha f64[2]) device views when
        the body runs inside / ahead of a hipGraph; None on the eager path."""
        cfg = self.cfg
        self._img_versions = None             # this path writes the parameters without the fused step's weight images
        states, actions, rewards, next_states, dones = self.memory.gather(indices)
        B = states.shape[0]
        self._sums.zero_()
        if eps_next is None and eps_cur is None:                               # both N(0,1) draws of the update in one launch
            eps_next, eps_cur = torch.randn(2, B, actions.shape[1], device=self.device)
        with torch.no_grad():                                                  # :233-237
            next_actions, next_logp = self.actor.sample(next_states, eps_next)
            tq1, tq2 = self.critic_target(next_states, next_actions)
            y = ops.sac_target(rewards, dones, tq1.view(-1), tq2.view(-1), next_logp.view(-1).contiguous(),
                               self.log_alpha, cfg.gamma)
        q1, q2 = self.critic(states, actions)                                  # :239-246
        dq1, dq2 = ops.sac_critic_loss(q1.view(-1), q2.view(-1), y, self._sums)
        self._critic_sink.arm()                                                # critic_optimizer.zero_grad()
        torch.autograd.backward([q1, q2], [dq1.view_as(q1), dq2.view_as(q2)])
        self._critic_sink.collect()
        # (+ the soft target update :265 of the critic just written, in the same launch: nothing reads the target before it)
        self.critic_optimizer.step(bias_dev=None if bias is None else bias[0], polyak=(self.critic_target_flat, cfg.tau))

        new_actions, logp = self.actor.sample(states, eps_cur)                 # :248-255
        with frozen_parameters(self.critic):                                   # its share of this backward is never computed
            q1, q2 = self.critic(states, new_actions)
        dlogp, dq1, dq2 = ops.sac_actor_loss(logp.view(-1).contiguous(), q1.view(-1), q2.view(-1), self.log_alpha,
                                             self.target_entropy, self._sums)
        self._actor_sink.arm()
        torch.autograd.backward([q1, q2, logp], [dq1.view_as(q1), dq2.view_as(q2), dlogp.view_as(logp)])
        self._actor_sink.collect()
        self.actor_optimizer.step(bias_dev=None if bias is None else bias[1])

        if bias is None:                                                       # :257-263
            self._alpha_steps += 1
        ops.sac_alpha_step(self.log_alpha, self._alpha_m, self._alpha_v, self._sums, B, cfg.lr_alpha,
                           max(self._alpha_steps, 1), loss_out=self._alpha_loss, bias_dev=None if bias is None else bias[2])
        return B

    def update_async(self):
        """The same update without a host round trip: one eager index draw + one scalar store, then the captured
        hipGraph of `_update_body` (gymrl_amd/graphs.py).  Losses stay on the device (`_sums`, `_alpha_loss`)."""
        cfg, m = self.cfg, self.memory
        if len(m) < cfg.batch_size:
            return
        if self._fused_update_ok():        # four launches: nothing left for a graph to save
            return self._update_fused()
        if self._graph is None:
            from .graphs import GraphedStep, StepScalars
            sc = self._scalars = StepScalars(self.device)
            (bc, self._off_c), (ba, self._off_a), (bl, self._off_l) = (sc.slot(16, torch.float32), sc.slot(16, torch.float32),
                                                                     sc.slot(16, torch.float64))
            if getattr(self, "_g_idx", None) is None:
                self._g_idx = torch.empty(cfg.batch_size, dtype=torch.int32, device=self.device)
            self._graph = GraphedStep(lambda: self._update_body(self._g_idx, bias=(bc, ba, bl)))
        m.draw_indices(cfg.batch_size, out=self._g_idx)
        sc = self._scalars
        sc.set(self._off_c, self.critic_optimizer.next_bias())
        sc.set(self._off_a, self.actor_optimizer.next_bias())
        self._alpha_steps += 1
        sc.set_doubles(self._off_l, 1.0 - 0.9 ** self._alpha_steps, 1.0 - 0.999 ** self._alpha_steps)
        sc.flush()
        self._graph()

    def save_checkpoint(self, path, include_memory=True):
        """ModelLoader-style dict: three networks, both optimisers in torch.optim.Adam's layout, the
        float64 temperature with its Adam state and — unlike the reference, which skips `memory` — the replay ring."""
        from .utils import checkpoint
        extra = {"memory_state_dict": self.memory.state_dict()} if include_memory else {}
        return checkpoint.save_agent(path, {"actor": self.actor, "critic": self.critic, "critic_target": self.critic_target},
                                     {"actor_optimizer": (self.actor, self.actor_optimizer),
                                      "critic_optimizer": (self.critic, self.critic_optimizer)},
                                     log_alpha=self.log_alpha.detach().cpu(), alpha_m=self._alpha_m.cpu(),
                                     alpha_v=self._alpha_v.cpu(), alpha_steps=self._alpha_steps,
                                     act_noise=self._act_noise, upd_noise=self._upd_noise,
                                     episode_rewards=list(self.episode_rewards), **extra)

    def load_checkpoint(self, path):
        from .utils import checkpoint
        rest = checkpoint.load_agent(path, {"actor": self.actor, "critic": self.critic, "critic_target": self.critic_target},
                                     {"actor_optimizer": (self.actor, self.actor_optimizer),
                                      "critic_optimizer": (self.critic, self.critic_optimizer)})
        self._img_versions = None             # the fused step's weight images are rebuilt from the loaded parameters
        self.log_alpha.copy_(rest["log_alpha"].to(self.device))
        self._alpha_m.copy_(rest["alpha_m"].to(self.device))
        self._alpha_v.copy_(rest["alpha_v"].to(self.device))
        self._alpha_steps = int(rest["alpha_steps"])
        self._act_noise, self._upd_noise = int(rest.get("act_noise", 0)), int(rest.get("upd_noise", 0))
        self.episode_rewards.clear()
        self.episode_rewards.extend(rest.get("episode_rewards", []))
        if "memory_state_dict" in rest:
            self.memory.load_state_dict(rest["memory_state_dict"])
        return rest

    def train(self, max_vector_steps=None):
        """The reference's train() loop (every Linear of the update and of acting is a gymrl_lin_* launch: gymrl_amd/nn.py)."""
        return self._train(max_vector_steps)

    CHUNK = 16     # vector steps per StepChunk replay (= the episode tracker's flush period)

    def _loop_buffers(self, N, D):
        """Step buffers that outlive one train() call: the captured StepChunk graph holds their addresses."""
        lb = getattr(self, "_loop", None)
        if lb is None or lb["N"] != N:
            d = self.device
            lb = self._loop = dict(N=N, obs=torch.empty(N, D, device=d), nxt=torch.empty(N, D, device=d),
                                   tobs=torch.empty(N, D, device=d), rew=torch.empty(N, device=d),
                                   tracker=EpisodeTracker(N, d, flush_every=1 if N == 1 else self.CHUNK))
        lb["tracker"].k, lb["tracker"].episodes = 0, 0
        return lb

    def _vector_step(self, lb, obs, nxt, ep_ret, done, cursor_dev=None, noise_dev=None, launch=True):
        """One vector step of :269-310 up to (not including) the update.  launch=False (fused path only): the acting launch's
        arguments are prepared and the caller issues them together with the update's (ops.sac_step)."""
        cfg, env = self.cfg, self.env
        if self._fused_ok():               # forward + draw + env step + ring row: one launch
            m = self.memory
            eps = None if self._parity_eps is None else next(self._parity_eps)
            if cursor_dev is None:
                self._act_noise += 1
            ops.sac_act_step(self._fused_args()[0], env, obs, nxt, cursor=m.cursor, cursor_dev=cursor_dev, eps=eps,
                             noise_seed=self.base_seed, noise_counter=self._act_noise, noise_counter_dev=noise_dev,
                             rew_out=lb["rew"], done_out=done, ep_ret_out=ep_ret, ep_stats=env.ep_stats, launch=launch)
            if cursor_dev is None:
                m.advance(env.n)
            return
        action = self.select_action(obs, eps=None if self._parity_eps is None else next(self._parity_eps))
        env.step(action, nxt, lb["rew"], done_out=done, term_obs_out=lb["tobs"], ep_ret_out=ep_ret)
        if cursor_dev is None:
            self.memory.push(obs, action, lb["rew"], lb["tobs"], done)           # done = terminated or truncated (:283)
        else:
            self.memory.push(obs, action, lb["rew"], lb["tobs"], done, cursor_dev=cursor_dev)
        if cfg.max_steps < env.max_steps:           # :278 `for step in range(cfg.max_steps)`: the episode is abandoned without a
            env.abandon(cfg.max_steps, nxt, done, ep_ret)         # done flag (stored above) and the next one starts

    def _chunk_body(self, lb, j):
        """Vector step j of a StepChunk capture: acting + env + ring append + index draw + update, every per-step scalar
        read from record j."""
        ch, tr = self._chunk, lb["tracker"]
        obs, nxt = (lb["obs"], lb["nxt"]) if j % 2 == 0 else (lb["nxt"], lb["obs"])
        if self._fused_ok():
            one = bool(getattr(self.cfg, "one_launch_step", False)) and self.cfg.batch_size <= 256   # acting + update as ONE launch (gymrl_sac_step: every waiting block resident)
            self._vector_step(lb, obs, nxt, tr.ret[j], tr.done[j], cursor_dev=ch.view(j, "push"), noise_dev=ch.view(j, "noise_a"),
                              launch=not one)
            self._update_fused(dev=(ch.view(j, "draw"), ch.view(j, "adam_c", torch.float32), ch.view(j, "adam_a", torch.float32),
                                    ch.view(j, "alpha", torch.float64), ch.view(j, "noise_u")), launch=not one)
            if one:
                ops.sac_step(*self._fused_args()[:2])
            return
        self._vector_step(lb, obs, nxt, tr.ret[j], tr.done[j], cursor_dev=ch.view(j, "push"))
        self.memory.draw_indices(self.cfg.batch_size, out=self._g_idx, dev=ch.view(j, "draw"))
        self._update_body(self._g_idx, bias=(ch.view(j, "adam_c", torch.float32), ch.view(j, "adam_a", torch.float32),
                                             ch.view(j, "alpha", torch.float64)))

    def _stage_chunk(self):
        """The host's bookkeeping of the next CHUNK vector steps, in the eager loop's order, written into the records."""
        ch, m, N = self._chunk, self.memory, self.env.n
        for j in range(ch.K):
            ch.set(j, "push", m.cursor)
            m.advance(N)
            ch.set(j, "draw", m.draws, m.size)
            m.draws += 1
            ch.set_bytes(j, "adam_c", self.critic_optimizer.next_bias())
            ch.set_bytes(j, "adam_a", self.actor_optimizer.next_bias())
            self._alpha_steps += 1
            ch.set(j, "alpha", 1.0 - 0.9 ** self._alpha_steps, 1.0 - 0.999 ** self._alpha_steps)
            self._act_noise += 1
            self._upd_noise += 1
            ch.set(j, "noise_a", self._act_noise)
            ch.set(j, "noise_u", self._upd_noise)
        ch.flush()

    def _train(self, max_vector_steps=None):
        """:269-310 with N lock-stepped envs.  With hipGraphs on, CHUNK whole vector steps (acting, env step, ring append,
        index draw, update) replay as one graph (gymrl_amd/graphs.py StepChunk)."""
        cfg, env = self.cfg, self.env
        N, D = env.n, env.obs_dim
        lb = self._loop_buffers(N, D)
        obs, nxt, tracker = lb["obs"], lb["nxt"], lb["tracker"]
        env.reset(obs)
        step = 0
        pending = None            # drain_async() token of the last chunk, collected one chunk later
        graphed = bool(getattr(cfg, "use_graphs", True)) and self._parity_updates is None
        # the fused step is 5 launches: 16 steps replay as ONE 80-node graph (the layer-by-layer step's 944-node chunk is slower
        # than its eager loop, hence chunk_steps = 0 there)
        want_chunk = getattr(cfg, "chunk_steps", self.CHUNK) > 0 or self._fused_ok()
        chunked = graphed and N > 1 and cfg.updates_per_step == 1 and want_chunk and self._parity_eps is None
        limit = max_vector_steps or (cfg.max_episodes * cfg.max_steps // N + 1)
        while tracker.episodes < cfg.max_episodes and step < limit:
            if (chunked and tracker.k == 0 and limit - step >= self.CHUNK and obs is lb["obs"]
                    and len(self.memory) >= cfg.batch_size):
                if getattr(self, "_chunk", None) is None:
                    from .graphs import StepChunk
                    self._chunk = StepChunk(self.device, self.CHUNK, [("push", "q"), ("draw", "Qq"), ("adam_c", "4f"),
                                                                      ("adam_a", "4f"), ("alpha", "2d"), ("noise_a", "Q"),
                                                                      ("noise_u", "Q")])
                    if getattr(self, "_g_idx", None) is None:
                        self._g_idx = torch.empty(cfg.batch_size, dtype=torch.int32, device=self.device)
                if self._fused_update_ok():
                    self._fused_args()         # weight images rebuilt (if stale) BEFORE the capture, not inside it
                self._stage_chunk()
                self._chunk.run(lambda j: self._chunk_body(lb, j), key=(id(env), env.state.data_ptr()))
                step += self.CHUNK
                tracker.k = self.CHUNK
                # the chunk's episode returns come back through a pinned buffer one chunk late: the host goes straight on
                # to staging the next chunk while this one runs (a sync here left the GPU idle for the ~0.5 ms of host work
                # per chunk: a quarter of a fused SAC step)
                token = tracker.drain_async()
                tracker.collect(pending, self.episode_rewards)
                pending = token
                if cfg.max_episodes - tracker.episodes <= N * self.CHUNK:      # within reach of the episode budget (:269): no lag,
                    tracker.collect(pending, self.episode_rewards)             # the loop stops where the eager loop would
                    pending = None
                continue
            tracker.collect(pending, self.episode_rewards)
            pending = None
            ep_ret, done = tracker.slot()
            self._vector_step(lb, obs, nxt, ep_ret, done)
            for _ in range(cfg.updates_per_step):
                if graphed:
                    self.update_async()
                else:
                    self.update()
            obs, nxt = nxt, obs
            step += 1
            tracker.advance(self.episode_rewards)
        tracker.collect(pending, self.episode_rewards)
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


if __name__ == "__main__":       # python -m gymrl_amd.sac_pendulum [--<Config attribute> <value> ...]  (sac_pendulum.py:354-370)
    from .utils.cli import run_script
    run_script(Config, SACTrainer)
