"""PPO (clip + dual-clip + GAE) — MI355X engine behind the reference's
algorithms/ppo_lunarlander.py surface: Config :29-52, ActorCritic :63-117,
RolloutBuffer :120-154, PPOTrainer :157-426 (compute_gae :179-196,
collect_rollout :198-231, update :233-330, train :332-366, eval :368-399).

Same class / method / attribute names and return types; what changed underneath:
  * `num_envs` env instances step in lock-step on the GPU (one per lane) instead of
    one gymnasium env; the rollout is a time-major SoA slab [T][N] in HBM;
  * categorical sampling, GAE (+ advantage moments), the clipped-surrogate loss
    forward/backward (+ metrics) and clip-norm + Adam are HIP kernels behind the
    C-ABI (include/gymrl.h), the update's GEMMs included (csrc/gemm.hip); PyTorch-ROCm owns memory and streams;
  * host syncs: one per rollout (episode returns) and one per update (metrics),
    instead of 3 per env step and 5 per minibatch;
  * with torch.distributed initialised every rank owns `num_envs` envs and the
    flat gradient is all-reduced (RCCL over xGMI) once per optimiser step.
"""
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from . import dist as gdist
from . import ops
from .envs import VecEnv
from .flat import FusedAdam, flatten_module
from . import ppo_net
from .nn import FusedMLP, Linear


class Config:
    def __init__(self):
        self.env_name = "LunarLander-v3"
        self.seed = None

        self.max_train_steps = 1_000_000
        self.update_freq = 2048          # steps PER ENV per rollout (T); the reference has one env
        self.num_epochs = 10
        self.batch_size = 64             # minibatch size when num_minibatches is None
        self.num_minibatches = None      # if set: minibatch = T*N / num_minibatches (reference ratio: 32)

        self.gamma = 0.99
        self.gae_lambda = 0.95
        self.clip_eps = 0.2
        self.dual_clip = 3.0
        self.entropy_coef = 0.01
        self.value_coef = 0.5
        self.max_grad_norm = 0.5

        self.lr = 3e-4
        self.anneal_lr = True

        self.hidden_dim = 256

        self.device = "cuda"             # MI355X only: there is no CPU path in this package
        # --- additions of the vectorised engine (defaults reproduce the reference loop) ---
        self.num_envs = 1
        self.scalar_surface = True       # num_envs == 1: collect_rollout() returns the python float of :231 (False: f32[1] on the device)
        self.reset_each_rollout = True   # ppo_lunarlander.py:200 resets the env at every rollout start
        self.gae_variant = 1             # 1 = time-blocked scan, 0 = sequential reference order
        self.gae_carry_in_rollout = True  # LunarLander's persistent rollout also runs the scan's carry pass at its tail (same bits; one launch fewer)
        self.fused_policy_forward = True  # rollout forward = one gymrl_mlp_forward launch (False: per-layer torch)
        self.fused_update = True          # update = ppo_net.FusedActorCriticUpdate.step(): hand-written f32-MFMA GEMMs + fused
                                          # HBM passes, hidden_dim 64 / 128 / 256 (False or another shape: torch autograd)
        self.persistent_rollout = True    # LunarLander: whole chunks of vector steps in one launch (gymrl_rollout_lunar)
        self.rollout_chunk = 0            # vector steps per persistent launch (0: the whole rollout in one launch —
                                          # every extra launch boundary waits for the slowest workgroup again)
        self.rollout_refill = True        # persistent rollout: wave 1 prepares every env's next episode while wave 0 steps
        self.use_graphs = False           # one rank: the minibatch body (gather, step(), norm + Adam) replayed as a hipGraph.
                                          # Bit-identical to the eager loop, but measured SLOWER at bench shape (update 889 vs 881
                                          # ms, A/B in one call: the host already runs ~19 launches ahead of 2.8 ms of kernels,
                                          # and the graph executor's 4-7 us per kernel node exceed the eager queue's gaps)
        self.solved_reward = 200.0


def layer_init(layer, std=np.sqrt(2)):
    """ppo_lunarlander.py:55-60."""
    if isinstance(layer, nn.Linear):
        nn.init.orthogonal_(layer.weight, gain=std)
        if layer.bias is not None:
            nn.init.constant_(layer.bias, 0)
    return layer


class ActorCritic(nn.Module):
    """Same architecture and init as ppo_lunarlander.py:63-90 (200,965 params at 8/4/256)."""

    def __init__(self, state_dim, action_dim, hidden_dim=256):
        super().__init__()
        # gymrl_amd.nn.Linear == nn.Linear with a split-K weight-gradient GEMM for huge minibatches
        self.shared = nn.Sequential(
            layer_init(Linear(state_dim, hidden_dim)), nn.Tanh(),
            layer_init(Linear(hidden_dim, hidden_dim)), nn.Tanh(),
        )
        self.actor = nn.Sequential(
            layer_init(Linear(hidden_dim, hidden_dim)), nn.Tanh(),
            layer_init(Linear(hidden_dim, action_dim), std=0.01),
        )
        self.critic = nn.Sequential(
            layer_init(Linear(hidden_dim, hidden_dim)), nn.Tanh(),
            layer_init(Linear(hidden_dim, 1), std=1.0),
        )
        self._fused = None

    def forward(self, x):
        features = self.shared(x)
        return self.actor(features), self.critic(features)

    def _act_net(self):
        if self._fused is None:
            # actor.0 / critic.0 (both read the trunk, disjoint outputs) and the two heads are independent pairs:
            # the kernel runs each pair in one barrier interval
            stages = [(self.shared[0], "tanh", -1, 0), (self.shared[2], "tanh", 0, 1),
                      (self.actor[0], "tanh", 1, 0), (self.critic[0], "tanh", 1, 2),
                      (self.actor[2], None, 0, -1), (self.critic[2], None, 2, -1)]
            # wider-than-LDS networks keep the per-layer library path
            ok = FusedMLP.supported(stages, self.shared[0].in_features)
            self._fused = FusedMLP(stages) if ok else False
        return self._fused

    def refresh_act(self):
        """Re-pack the weights act_forward(refresh=False) reads (call after every parameter update)."""
        if self._act_net():
            self._fused.refresh()

    @torch.no_grad()
    def act_forward(self, x, refresh=True):
        """Inference forward for the rollout (:92-108): the six Linear(+Tanh) layers in one launch
        (gymrl_mlp_forward).  Returns (logits [N, A], value [N, 1]); the tensors are reused by the
        next call with the same batch size.  refresh=False skips re-packing the weights (the
        rollout loop packs once at its start: parameters do not change inside a rollout)."""
        net = self._act_net()
        if net is False:
            return self.forward(x)
        logits, value = net(x, refresh)
        return logits, value

    @torch.no_grad()
    def get_action(self, state, deterministic=False, seed=0, counter=0, env_id0=0, noise_exp=None):
        """Batched counterpart of :92-104: state [N, obs] -> (action i32[N], logp[N], value[N])."""
        logits, value = self.forward(state)
        act, logp, _, val = ops.categorical_sample(logits, value=value.squeeze(-1).contiguous(),
                                                   noise_exp=noise_exp, seed=seed, counter=counter,
                                                   env_id0=env_id0, deterministic=deterministic)
        return act, logp, val

    @torch.no_grad()
    def get_value(self, state):
        return self.forward(state)[1].squeeze(-1)


class RolloutBuffer:
    """Time-major SoA slab [T][N] in HBM replacing the six python lists of :120-154."""

    def __init__(self, T, N, obs_dim, device):
        self.T, self.N = T, N
        self.states = torch.zeros(T + 1, N, obs_dim, device=device)      # row T = bootstrap observation
        self.actions = torch.zeros(T, N, dtype=torch.int32, device=device)
        self.log_probs = torch.zeros(T, N, device=device)
        self.values = torch.zeros(T, N, device=device)
        self.rewards = torch.zeros(T, N, device=device)
        self.dones = torch.zeros(T, N, dtype=torch.uint8, device=device)
        self.ep_returns = torch.zeros(T, N, device=device)                # valid where dones == 1
        self.advantages = torch.zeros(T, N, device=device)
        self.returns = torch.zeros(T, N, device=device)
        self.pos = 0

    def clear(self):
        self.pos = 0

    def __len__(self):
        return self.pos * self.N


class KernelTimers:
    """HIP-event timers on torch's current stream (the stream the C-ABI launches on):
    per label, summed device time and work units — bench.py turns them into achieved
    GB/s / TFLOP/s against the roofline.  `every` > 1 samples: the update brackets the launches of every
    `every`-th minibatch only (tick()), so that the instrumentation — two event records around each of the ~19 launches of
    a minibatch — does not itself stretch the region it measures; launches outside the minibatch loop (rollout, GAE)
    are always bracketed."""

    def __init__(self, every=1):
        self.pairs = {}
        self._open = {}
        self.every, self._n, self.live = max(1, int(every)), 0, True

    def tick(self):
        """Start of a minibatch: decides whether its launches are bracketed."""
        self.live = self._n % self.every == 0
        self._n += 1

    def resume(self):
        """End of the minibatch loop: everything is bracketed again."""
        self.live = True

    def start(self, label):
        if not self.live:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._open[label] = e

    def stop(self, label, units):
        if not self.live:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.pairs.setdefault(label, []).append((self._open.pop(label), e, units))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for label, prs in self.pairs.items():
            ms = [s.elapsed_time(e) for s, e, _ in prs]
            out[label] = dict(launches=len(prs), total_s=sum(ms) * 1e-3, units=sum(u for _, _, u in prs),
                              avg_us=1e3 * sum(ms) / len(ms))
        return out

    def reset(self):
        self.pairs.clear()


class PPOTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd.PPOTrainer needs an MI355X (gfx950) and libgymrl_hip.so; no CPU fallback")
        self.rank, self.world_size = gdist.rank(), gdist.world_size()
        self.collective = gdist.collectives_active()          # world_size > 1 (or one rank under GYMRL_FORCE_COLLECTIVES)
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        N = int(config.num_envs)
        self.env = VecEnv(config.env_name, N, device=self.device, seed=self.base_seed, env_id0=self.rank * N)

        state_dim = self.env.observation_space.shape[0]
        action_dim = self.env.action_space.n
        self.action_dim = action_dim

        # identical initial parameters on every rank: CPU init under a fixed torch seed
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.model = ActorCritic(state_dim, action_dim, config.hidden_dim)
        torch.random.set_rng_state(gen_state)
        self.flat_params, self.flat_grads = flatten_module(self.model, self.device, order=ppo_net.LAYOUT)
        self._fused_update = None      # built on first update() when cfg.fused_update and the shape allows it
        gdist.broadcast(self.flat_params)
        self.optimizer = FusedAdam(self.flat_params, self.flat_grads, lr=config.lr, eps=1e-5,
                                   max_grad_norm=config.max_grad_norm)

        T = int(config.update_freq)
        self.buffer = RolloutBuffer(T, N, state_dim, self.device)
        self.step_count = 0
        self.rollout_count = 0
        self.episode_rewards = deque(maxlen=100)
        self._gae_ws = ops.gae_workspace(T, N, self.device)
        self._moments = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._next_value = torch.zeros(N, device=self.device)
        self._perm_seed, self._perm_draws, self._perm = self.base_seed * 7919 + 13 + self.rank, 0, None   # epoch shuffles
        self._loss_cfg = (config.clip_eps, config.dual_clip, config.value_coef, config.entropy_coef)
        self._finished = None
        self._metric_parts = None
        self._agg_ready = self._carry_ready = False
        self._parity_noise = None      # tests: f32[rollouts, T, N, A] Exp(1) draws (ops.categorical_sample noise_exp)
        self._parity_indices = []      # tests: per-update [num_epochs, T*N] shuffle orders consumed by update()
        self._wg_ticks = None          # profiling: i64[2 * ceil(N/16)] start/end ticks of the last persistent launch
        self._eval_env_factory = None  # tests: num_episodes -> env object for eval() (default: a fresh VecEnv)
        self._gae_running = torch.zeros(2, N, dtype=torch.float64, device=self.device)
        self._packed = None      # [T*N, 16] packed rollout records (allocated on first update)
        mb = self._minibatch_size_for(T * N)
        self._stage = (torch.empty(mb, state_dim, device=self.device), torch.empty(mb, dtype=torch.int32, device=self.device),
                       torch.empty(mb, device=self.device), torch.empty(mb, device=self.device),
                       torch.empty(mb, device=self.device))
        self._stage2 = tuple(torch.empty_like(t) for t in self._stage) if self.collective else None
        self._reducer = None     # gdist.GradReducer over the flat gradient (world_size > 1; built on the first update)
        self._graph = None       # captured minibatch body (one rank; update())
        self._timers = None      # set to a KernelTimers() to time kernels with HIP events (bench.py)
        if self.rank == 0:
            print(f"Device: {self.device} x{self.world_size} | envs/GPU: {N} | rollout T: {T}")
            print(f"State dim: {state_dim}, Action dim: {action_dim}")
            print(f"Model parameters: {sum(p.numel() for p in self.model.parameters()):,}")

    # ------------------------------------------------------------------ GAE --
    def compute_gae(self, next_value=None):
        """:179-196 over every env of the slab; returns (advantages, returns) [T, N]
        (un-normalised; the whole-rollout moments are left in self._moments).  With
        next_value=None the bootstrap values of the last collect_rollout() are used.
        The reference's scalar surface (num_envs == 1): a python float / numpy bootstrap value in -> two float64
        np.ndarray [T] out, as `compute_gae(next_value) -> (advantages, returns)` at :179-196."""
        if next_value is not None and not torch.is_tensor(next_value):
            nv = torch.as_tensor(np.asarray(next_value, np.float32).reshape(-1), device=self.device).expand(self.buffer.N).contiguous()
            adv, ret = self._compute_gae(nv)
            return adv[:, 0].double().cpu().numpy(), ret[:, 0].double().cpu().numpy()
        return self._compute_gae(next_value)

    def _compute_gae(self, next_value=None):
        b = self.buffer
        nv = self._next_value if next_value is None else next_value
        # variant 2 = the blocked scan minus its first pass: the rollout composed the chunk maps
        variant = 2 if (self.cfg.gae_variant == 1 and self._agg_ready and next_value is None) else self.cfg.gae_variant
        if variant == 2 and getattr(self, "_carry_ready", False):
            variant = 3                  # ... and its second: the persistent rollout left the carries in the workspace too
        self._agg_ready = self._carry_ready = False
        self._last_gae_variant = variant
        if self._timers is not None:
            self._timers.start("gae")
        ops.gae(b.rewards, b.values, b.dones, nv, self.cfg.gamma, self.cfg.gae_lambda, b.advantages,
                b.returns, self._moments, variant, self._gae_ws)
        if self._timers is not None:
            self._timers.stop("gae", b.T * b.N)
        return b.advantages, b.returns

    # -------------------------------------------------------------- rollout --
    def collect_rollout(self):
        """:198-231 for N envs at once.  Returns next_value f32[N] (bootstrap) on the device — or, on the reference's
        scalar surface (num_envs == 1 and cfg.scalar_surface), the python float `collect_rollout() -> next_value` of :231;
        update() / compute_gae() take either."""
        nv = self._collect_rollout()
        if self.buffer.N == 1 and getattr(self.cfg, "scalar_surface", True):
            self._nv_float = float(nv[0])
            return self._nv_float
        return nv

    @torch.no_grad()
    def _collect_rollout(self):
        cfg, b, env = self.cfg, self.buffer, self.env
        b.clear()
        if cfg.reset_each_rollout or self.rollout_count == 0:
            # :200 env.reset(seed=cfg.seed): a fixed cfg.seed replays the same stream every
            # rollout (reference behaviour); seed=None draws a fresh stream per rollout.
            seed = self.base_seed if cfg.seed is not None else self.base_seed + 0x9E3779B1 * (self.rollout_count + 1)
            env.reset(b.states[0], seed=seed & 0x7FFFFFFFFFFFFFFF)
        else:
            b.states[0].copy_(b.states[b.T])
        counter0 = self.rollout_count * b.T
        tm = self._timers
        fuse_gae = cfg.gae_variant == 1 and b.N % 4 == 0
        # parity mode: explicit Exp(1) draws f32[rollouts, T, N, A] replace the Philox stream
        noise = None if self._parity_noise is None else self._parity_noise[self.rollout_count]
        if self._persistent_ok():
            return self._collect_rollout_persistent(counter0, noise, fuse_gae)
        if cfg.fused_policy_forward:
            self.model.refresh_act()                     # pack the current weights once per rollout
            fwd = lambda o: self.model.act_forward(o, refresh=False)   # noqa: E731
        else:
            fwd = self.model
        for t in range(b.T):
            logits, value = fwd(b.states[t])
            if tm is not None and t % 64 == 0:
                tm.start("env_step+sample")
            # while sampling step t, fold step t-1 (whose delta needs V_t) into its GAE chunk map
            online = (ops.gae_online(b.rewards[t - 1], b.dones[t - 1], b.values[t - 1], self._gae_running,
                                     self._gae_ws, t - 1, b.T, cfg.gamma, cfg.gae_lambda) if fuse_gae and t > 0 else None)
            ops.categorical_sample(logits, value=value.view(-1), noise_exp=None if noise is None else noise[t],
                                   seed=env.seed, counter=counter0 + t,
                                   env_id0=env.env_id0, act_out=b.actions[t], logp_out=b.log_probs[t],
                                   ent_out=None, value_out=b.values[t], online=online)
            env.step(b.actions[t], b.states[t + 1], b.rewards[t], done_out=b.dones[t],
                     ep_ret_out=b.ep_returns[t])
            if tm is not None and t % 64 == 0:
                tm.stop("env_step+sample", b.N)
        b.pos = b.T
        self.step_count += b.T * b.N
        self.rollout_count += 1
        self._next_value.copy_(fwd(b.states[b.T])[1].view(-1))
        if fuse_gae:
            ops.gae_online_flush(ops.gae_online(b.rewards[b.T - 1], b.dones[b.T - 1], b.values[b.T - 1],
                                                self._gae_running, self._gae_ws, b.T - 1, b.T, cfg.gamma,
                                                cfg.gae_lambda), self._next_value)
            self._agg_ready = True
        self._carry_ready = False        # (the step-by-step path composes the maps only: gymrl_gae variant 2 runs the carry pass)
        # :220-221 episode_rewards.append on done: compacted on the device here, read back by
        # _drain_episode_returns() once the update's kernels are queued (no sync in the rollout)
        self._finished = True
        return self._next_value

    def _persistent_ok(self):
        """gymrl_rollout_lunar / gymrl_rollout_cartpole cover: the built-in LunarLander and CartPole envs, a policy the
        one-launch forward supports."""
        cfg, env = self.cfg, self.env
        return (cfg.persistent_rollout and cfg.fused_policy_forward and isinstance(env, VecEnv)
                and (env.kind, self.action_dim) in ((ops.LUNARLANDER, 4), (ops.CARTPOLE, 2)) and bool(self.model._act_net()))

    def _collect_rollout_persistent(self, counter0, noise, fuse_gae):
        """:198-231 as ceil(T / rollout_chunk) launches: every workgroup runs its 16 envs through the whole chunk
        (policy forward, draw, GAE chunk maps, env step + reset-on-done, slab writes) without waiting for the others."""
        cfg, b, env = self.cfg, self.buffer, self.env
        net = self.model._act_net()
        net.refresh()                                  # pack the current weights once per rollout
        desc = net.descriptor(b.N, self.device)
        tm = self._timers
        chunk = int(cfg.rollout_chunk) if int(cfg.rollout_chunk) > 0 else b.T
        carry = bool(fuse_gae and getattr(cfg, "gae_carry_in_rollout", True) and env.kind == ops.LUNARLANDER)
        for t0 in range(0, b.T, chunk):
            n = min(chunk, b.T - t0)
            if tm is not None:
                tm.start("rollout_chunk")
            common = (env.state, b.N, env.seed, env.env_id0, counter0, b.states, b.actions, b.log_probs, b.values, b.rewards,
                      b.dones, b.ep_returns, self._next_value, desc, b.T, t0, n, cfg.gamma, cfg.gae_lambda)
            online = dict(noise_exp=noise, gae_running=self._gae_running if fuse_gae else None,
                          gae_workspace=self._gae_ws if fuse_gae else None, ep_stats=env.ep_stats)
            if env.kind == ops.CARTPOLE:
                ops.rollout_cartpole(*common, **online)
            else:
                ops.rollout_lunar(*common, **online, wg_ticks=self._wg_ticks, refill=getattr(cfg, "rollout_refill", True),
                                  gae_carry=carry)
            if tm is not None:
                tm.stop("rollout_chunk", n * b.N)
        b.pos = b.T
        self.step_count += b.T * b.N
        self.rollout_count += 1
        self._agg_ready = bool(fuse_gae)
        self._carry_ready = carry
        self._finished = True
        return self._next_value

    def _drain_episode_returns(self):
        if self._finished:
            b = self.buffer
            done = b.ep_returns[b.dones.bool()][-self.episode_rewards.maxlen:]   # (t, n) order == time order
            for r in done.tolist():
                self.episode_rewards.append(r)
            self._finished = None

    # --------------------------------------------------------------- update --
    def _minibatch_size_for(self, total):
        if self.cfg.num_minibatches:
            return max(1, total // int(self.cfg.num_minibatches))
        return min(int(self.cfg.batch_size), total)

    def _minibatch_size(self):
        return self._minibatch_size_for(len(self.buffer))

    def update(self, next_value=None, indices=None):
        """:233-330.  (next_value is accepted for signature compatibility; when it is the tensor
        collect_rollout() returned, the GAE maps composed during the rollout are reused.)  `indices` (optional i32/i64 [num_epochs, T*N]) replays an explicit
        shuffle order (parity mode); by default one keyed device permutation per epoch (gymrl_permutation, :262)."""
        cfg, b = self.cfg, self.buffer
        if indices is None and self._parity_indices:
            indices = self._parity_indices.pop(0)
        if next_value is not None and not torch.is_tensor(next_value):
            # scalar surface: the float collect_rollout() handed out IS the device value (f32 -> f64 is exact)
            next_value = (None if float(next_value) == getattr(self, "_nv_float", None)
                          else torch.full((b.N,), float(next_value), device=self.device))
        self._compute_gae(None if next_value is self._next_value else next_value)
        if self.collective:
            if self._timers is not None:
                self._timers.start("moments_allreduce")
            gdist.all_reduce_sum(self._moments)          # :236 mean/std over the WHOLE rollout (all ranks)
            if self._timers is not None:
                self._timers.stop("moments_allreduce", 3)
        total = len(b)
        mb = self._minibatch_size()
        n_mb = (total + mb - 1) // mb
        obs_dim = b.states.shape[-1]
        tm = self._timers
        # P6: one 64-B record per transition so that a shuffled minibatch is one random line per sample
        self._packed = ops.pack_rollout(b.states[:b.T].reshape(total, obs_dim), b.actions.view(-1),
                                        b.log_probs.view(-1), b.advantages.view(-1), b.returns.view(-1),
                                        self._packed)
        fu = None
        if cfg.fused_update and ppo_net.supported(self.model):
            if self._fused_update is None or self._fused_update.R < mb:
                self._fused_update = ppo_net.FusedActorCriticUpdate(self.model, mb)
            fu = self._fused_update
            fu.timers = tm
        one_pass = fu is not None
        # block partials of every minibatch's 5 metric sums; reduced by ONE launch after the last step
        nblk = fu.metric_blocks(mb) if one_pass else ops.loss_blocks(mb)
        if self._metric_parts is None or self._metric_parts.shape[:2] != (cfg.num_epochs * n_mb, nblk):
            self._metric_parts = torch.zeros(cfg.num_epochs * n_mb, nblk, 5, dtype=torch.float64, device=self.device)
        else:
            self._metric_parts.zero_()
        red = None
        if self.collective:
            # the flat gradient is all-reduced in two buckets on a communication stream.  The
            # tail of the flat buffer (actor.0 | critic.0 and the heads: 2/3 of the bytes) is complete once the N = 512
            # weight gradient is queued and is reduced under the trunk's three GEMMs; the head follows the last backward
            # kernel and overlaps the next minibatch's gather.
            if self._reducer is None:
                split = (self.model.actor[0].weight.data_ptr() - self.flat_params.data_ptr()) // 4
                self._reducer = gdist.GradReducer(self.flat_grads, [split] if one_pass else None)
            red = self._reducer
            red.timed = tm is not None
        stages = (self._stage, self._stage2) if red is not None else (self._stage, self._stage)

        def minibatches():
            for epoch in range(cfg.num_epochs):
                if indices is not None:
                    perm = torch.as_tensor(indices[epoch], device=self.device).to(torch.int32)
                else:
                    self._perm_draws += 1
                    self._perm = ops.permutation(self._perm_seed, self._perm_draws, total, self.device,
                                                 out=self._perm if self._perm is not None and self._perm.numel() == total else None)
                    perm = self._perm
                for start in range(0, total, mb):
                    yield perm[start:start + mb]

        def gather(mb_idx, k):
            B = mb_idx.numel()
            if tm is not None:
                tm.start("gather_minibatch")
            out = ops.gather_minibatch(self._packed, mb_idx, obs_dim, stages[k & 1] if B == mb else None)
            if tm is not None:
                tm.stop("gather_minibatch", B)
            return out

        # One rank, full-size minibatches: the minibatch body (gather, the ~17 launches of step(), norm + Adam) is captured once
        # as a hipGraph and replayed; per minibatch the host issues an index copy, one scalar store (Adam's bias corrections
        # and learning rate travel through device memory), the replay, and the copy of the metric row.  Same kernels, same
        # order: bit-identical to the eager loop (tests/test_graphs_gpu.py).  Minibatches whose launches bench.py's timers
        # bracket run eagerly (events cannot be recorded into a replay).
        graphed = (one_pass and red is None and bool(getattr(cfg, "use_graphs", False)) and total % mb == 0
                   and cfg.num_epochs * n_mb > 3)
        gs = self._graph_state(fu, mb, obs_dim) if graphed else None

        sizes = []
        row = 0
        it = minibatches()
        first = next(it)
        cur = gather(first, 0) if not graphed else first
        while cur is not None:
            if tm is not None:
                tm.tick()                            # (sampling timers: this minibatch's launches are bracketed or not)
            if graphed:
                mb_idx = cur
                B = mb_idx.numel()
                if tm is not None and tm.live:       # a bracketed minibatch: the eager sequence (and its timers)
                    fu.timers = tm
                    g5 = gather(mb_idx, 0)
                    fu.step(*g5, self._loss_cfg, self._moments, self._metric_parts[row])
                    tm.start("adam_step")
                    self.optimizer.step(grad_scale=1.0)
                    tm.stop("adam_step", self.flat_params.numel())
                else:
                    fu.timers = None
                    gs["idx"].copy_(mb_idx)
                    gs["scalars"].set(gs["off"], self.optimizer.next_bias())
                    gs["scalars"].flush()
                    gs["step"]()                     # two eager warm-ups per trainer, then capture + replay
                    self._metric_parts[row].copy_(gs["metric"])
                sizes.append(B)
                row += 1
                cur = next(it, None)
                continue
            mb_obs, mb_act, mb_lp, mb_adv, mb_ret = cur
            B = mb_obs.shape[0]
            if one_pass:
                fu.step(mb_obs, mb_act, mb_lp, mb_adv, mb_ret, self._loss_cfg, self._moments, self._metric_parts[row],
                        reducer=red)
            else:                                     # shapes ppo_net does not cover (or cfg.fused_update = False): autograd
                logits, values = self.model(mb_obs)
                values = values.view(-1)
                dlogits = torch.empty_like(logits)
                dvalues = torch.empty_like(values)
                if tm is not None:
                    tm.start("ppo_loss_fwd_bwd")
                ops.ppo_loss_fwd_bwd(logits, values, mb_act, mb_lp, mb_adv, mb_ret, self._loss_cfg,
                                     adv_moments=self._moments, dlogits_out=dlogits, dvalue_out=dvalues,
                                     workspace=self._metric_parts[row])
                if tm is not None:
                    tm.stop("ppo_loss_fwd_bwd", B)
                torch.autograd.backward([logits, values], [dlogits, dvalues])
                if red is not None:
                    red.launch(0)
            # the next minibatch's rows do not depend on the parameters: gathered while the last bucket is in flight
            nxt = next(it, None)
            cur = gather(nxt, row + 1) if nxt is not None else None
            if red is not None:
                red.wait()
            if tm is not None:
                tm.start("adam_step")
            self.optimizer.step(grad_scale=1.0 / self.world_size)
            if tm is not None:
                tm.stop("adam_step", self.flat_params.numel())
            sizes.append(B)
            row += 1
        if tm is not None:
            tm.resume()
        metrics = ops.reduce_rows(self._metric_parts, row, nblk, 5)
        self._drain_episode_returns()
        m = metrics.cpu().numpy() / np.asarray(sizes, np.float64)[:, None]   # the one host sync of the update
        m = m.mean(axis=0)
        return {"policy_loss": m[0], "value_loss": m[1], "entropy": m[2], "clip_frac": m[3], "approx_kl": m[4]}

    def _graph_state(self, fu, mb, obs_dim):
        """Fixed buffers + the captured minibatch body (rebuilt when the update object, the minibatch size or the packed
        rollout's address change)."""
        key = (id(fu), mb, self._packed.data_ptr(), self._stage[0].data_ptr())
        gs = self._graph
        if gs is None or gs["key"] != key:
            from .graphs import GraphedStep, StepScalars
            sc = StepScalars(self.device)
            bias, off = sc.slot(16, torch.float32)
            gs = self._graph = dict(key=key, scalars=sc, bias=bias, off=off,
                                    idx=torch.empty(mb, dtype=torch.int32, device=self.device),
                                    metric=torch.zeros(fu.metric_blocks(mb), 5, dtype=torch.float64, device=self.device))

            def body():
                g5 = ops.gather_minibatch(self._packed, gs["idx"], obs_dim, self._stage)
                fu.step(*g5, self._loss_cfg, self._moments, gs["metric"])
                self.optimizer.step(grad_scale=1.0, bias_dev=gs["bias"])
            gs["step"] = GraphedStep(body, warmup=2)
        return gs

    # ----------------------------------------------------------- checkpoint --
    def save_checkpoint(self, path):
        """ModelLoader-style dict (utils/model.py:337-349): `model_state_dict`, `optimizer_state_dict`
        (torch.optim.Adam layout) + the counters a resumed run needs."""
        from .utils import checkpoint
        return checkpoint.save_agent(path, {"model": self.model}, {"optimizer": (self.model, self.optimizer)},
                                     step_count=self.step_count, rollout_count=self.rollout_count,
                                     perm_draws=self._perm_draws, episode_rewards=list(self.episode_rewards))

    def load_checkpoint(self, path):
        from .utils import checkpoint
        rest = checkpoint.load_agent(path, {"model": self.model}, {"optimizer": (self.model, self.optimizer)})
        self.step_count = int(rest.get("step_count", 0))
        self.rollout_count = int(rest.get("rollout_count", 0))
        self._perm_draws = int(rest.get("perm_draws", 0))
        self.episode_rewards.clear()
        self.episode_rewards.extend(rest.get("episode_rewards", []))
        gdist.broadcast(self.flat_params)
        return rest

    # ---------------------------------------------------------------- train --
    def train(self):
        if self.rank == 0:
            print("Starting training...")
        update_count = 0
        while self.step_count * self.world_size < self.cfg.max_train_steps:
            if self.cfg.anneal_lr:
                frac = 1.0 - self.step_count * self.world_size / self.cfg.max_train_steps
                lr = self.cfg.lr * frac
                for param_group in self.optimizer.param_groups:
                    param_group["lr"] = lr
            next_value = self.collect_rollout()
            metrics = self.update(next_value)
            update_count += 1
            if len(self.episode_rewards) > 0 and self.rank == 0:
                avg_reward = np.mean(self.episode_rewards)
                print(f"Step: {self.step_count * self.world_size:,} | Updates: {update_count} | "
                      f"Avg Reward: {avg_reward:.1f} | Policy Loss: {metrics['policy_loss']:.4f} | "
                      f"Value Loss: {metrics['value_loss']:.4f} | Entropy: {metrics['entropy']:.4f} | "
                      f"KL: {metrics['approx_kl']:.4f} | Clip: {metrics['clip_frac']:.2%}")
            # the stop decision is collective: every rank sees different episodes, and a rank that left alone would
            # strand the others in the next all-reduce
            solved = len(self.episode_rewards) >= 100 and np.mean(self.episode_rewards) >= self.cfg.solved_reward
            if self.collective:
                flag = torch.tensor([1.0 if solved else 0.0], device=self.device)
                gdist.all_reduce_max(flag)
                solved = bool(flag.item() > 0)
            if solved:
                if self.rank == 0:
                    print(f"\nEnvironment solved at step {self.step_count * self.world_size:,}!")
                break
        if self.rank == 0:
            print("Training completed!")
        self.env.close()

    @torch.no_grad()
    def eval(self, num_episodes=10):
        """:368-399: deterministic (argmax) episodes — run as `num_episodes` parallel envs,
        each contributing its first finished episode."""
        if self._eval_env_factory is not None:
            env = self._eval_env_factory(num_episodes)
        else:
            env = VecEnv(self.cfg.env_name, num_episodes, device=self.device, seed=self.base_seed + 1_000_003,
                         env_id0=1 << 40)
        obs = env.reset()
        nxt = torch.empty_like(obs)
        rew = torch.empty(num_episodes, device=self.device)
        done = torch.zeros(num_episodes, dtype=torch.uint8, device=self.device)
        ep_ret = torch.zeros(num_episodes, device=self.device)
        result = torch.full((num_episodes,), float("nan"), device=self.device)
        self.model.eval()
        for _ in range(env.max_steps + 1):
            act, _, _ = self.model.get_action(obs, deterministic=True)
            env.step(act, nxt, rew, done_out=done, ep_ret_out=ep_ret)
            first = done.bool() & torch.isnan(result)
            result = torch.where(first, ep_ret, result)
            obs, nxt = nxt, obs
            if not torch.isnan(result).any():
                break
        self.model.train()
        rewards = result.tolist()
        if self.rank == 0:
            print(f"Evaluation Results: Mean = {np.mean(rewards):.1f} +/- {np.std(rewards):.1f}")
        return rewards

    def test(self):
        """:401-426 without the render window (the batched env has no renderer)."""
        return self.eval(num_episodes=5)


if __name__ == "__main__":       # python -m gymrl_amd.ppo_lunarlander [--<Config attribute> <value> ...]  (ppo_lunarlander.py:429-445)
    from .utils.cli import run_script
    run_script(Config, PPOTrainer)
