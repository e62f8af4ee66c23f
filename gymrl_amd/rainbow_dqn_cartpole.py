"""Rainbow DQN (Double + Dueling + NoisyNet + PER sum-tree + n-step + soft target) — MI355X
engine behind the reference's algorithms/rainbow_dqn_cartpole.py surface: Config :32-48,
NoisyLinear :51-97, DuelingNoisyNetwork :100-113, SumTree :116-152,
PrioritizedNStepBuffer :155-264, RainbowDQNTrainer :267-446 (select_action :293-309,
update :311-361, train :363-405).

Underneath: per-env n-step windows + ring append, the float64 sum-tree (ordered batched
update, stratified sampling, priority_max), NoisyNet noise generation, the double-DQN TD
loss forward/backward, clip-norm + Adam and the soft target update are HIP kernels behind
the C-ABI; the four Linear layers run through PyTorch-ROCm autograd.
"""
import os
import copy
import math
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from . import nn as gnn
from . import ops
from .envs import EpisodeTracker, VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .nn import SmallLinear, small_linear
from .utils import scalar


class Config:
    def __init__(self):
        self.env_name = "CartPole-v1"
        self.seed = None
        self.max_episodes = 500
        self.max_steps = 500
        self.batch_size = 256
        self.gamma = 0.9
        self.tau = 0.005
        self.lr = 1e-3
        self.memory_capacity = 20000
        self.hidden_dim = 256
        self.n_steps = 5
        self.alpha = 0.6
        self.beta_init = 0.4
        self.grad_clip = 10.0
        self.device = "cuda"
        # --- vectorised-engine additions ---
        self.num_envs = 1
        self.updates_per_step = 1
        self.use_graphs = True             # replay the update as one captured hipGraph (train(); update() stays eager)
        self.fused_step = True             # acting + env + n-step push in ONE launch, the update's ~20 Linear / loss launches in TWO
        #                                    (csrc/offpolicy_step.hip; bit-identical to the layer-by-layer path, which remains for
        #                                    other shapes and custom envs)
        self.fused_images = True           # fused step: fc2 (hidden x hidden, hidden % 16 == 0) is streamed from MFMA-operand images
        #                                    that the step's NoisyLinear launch rebuilds from the parameters (same values, same
        #                                    registers: bit-identical; False: nn.Linear's rows are read in place)
        self.chunk_steps = 16              # whole vector steps (acting, env, n-step store, sum tree, draw, update) as one
        #                                    hipGraph per 16 (graphs.StepChunk); 0: eager acting + a graph per update.  The
        #                                    eager launches of a step cost the host 0.46 ms at N = 8192 — more than the GPU needs


class NoisyLinear(nn.Module):
    """rainbow_dqn_cartpole.py:51-97; the factorised noise is produced by gymrl_noisy_noise
    (in-kernel Box-Muller on Philox, or explicit raw N(0,1) draws in parity mode)."""

    _counter = 0

    def __init__(self, in_features, out_features, sigma_init=0.5, seed=0):
        super().__init__()
        self.in_features, self.out_features, self.sigma_init, self.seed = in_features, out_features, sigma_init, seed
        self.weight_mu = nn.Parameter(torch.empty(out_features, in_features))
        self.weight_sigma = nn.Parameter(torch.empty(out_features, in_features))
        self.register_buffer("weight_epsilon", torch.zeros(out_features, in_features))
        self.bias_mu = nn.Parameter(torch.empty(out_features))
        self.bias_sigma = nn.Parameter(torch.empty(out_features))
        self.register_buffer("bias_epsilon", torch.zeros(out_features))
        self.raw_noise = None            # parity mode: iterator of (eps_in_raw, eps_out_raw) device tensors
        self.dev_counters = None         # StepChunk capture: iterator of device uint64[1] draw counters (no host counter)
        self.staged = None               # hipGraph mode: [(w_eps, b_eps), ...] drawn ahead of the replay, consumed in order
        self._staged_k = 0
        self.reset_parameters()

    def reset_parameters(self):
        mu_range = 1 / math.sqrt(self.in_features)
        self.weight_mu.data.uniform_(-mu_range, mu_range)
        self.bias_mu.data.uniform_(-mu_range, mu_range)
        self.weight_sigma.data.fill_(self.sigma_init / math.sqrt(self.in_features))
        self.bias_sigma.data.fill_(self.sigma_init / math.sqrt(self.out_features))

    def draw_staged_at(self, counters):
        """hipGraph mode: draw the noise of the next len(counters) training-mode forwards with the given Philox
        counters (these counter-keyed launches stay outside the graph); reset_noise() then copies them in, in
        order, inside the graph."""
        if self.staged is None:
            self.staged = [(torch.zeros_like(self.weight_epsilon), torch.zeros_like(self.bias_epsilon)) for _ in counters]
        for (w_eps, b_eps), c in zip(self.staged, counters):
            ops.noisy_noise(self.in_features, self.out_features, w_eps, b_eps, seed=self.seed, counter=c)
        self._staged_k = 0

    def reset_noise(self):
        if not self.weight_epsilon.is_cuda:
            return                        # CPU construction time: the buffers are filled on first GPU forward
        if self.dev_counters is not None:
            ops.noisy_noise(self.in_features, self.out_features, self.weight_epsilon, self.bias_epsilon, seed=self.seed,
                            counter_dev=next(self.dev_counters))
        elif self.staged is not None and self._staged_k < len(self.staged):
            w_eps, b_eps = self.staged[self._staged_k]
            self._staged_k += 1
            self.weight_epsilon.copy_(w_eps)
            self.bias_epsilon.copy_(b_eps)
        elif self.raw_noise is not None:
            ei, eo = next(self.raw_noise)
            ops.noisy_noise(self.in_features, self.out_features, self.weight_epsilon, self.bias_epsilon, ei, eo)
        else:
            NoisyLinear._counter += 1
            ops.noisy_noise(self.in_features, self.out_features, self.weight_epsilon, self.bias_epsilon,
                            seed=self.seed, counter=NoisyLinear._counter)

    def noise_source(self):
        """Where the next training-mode forward's noise comes from, as gymrl_noisy_combine fields — the fused head's
        version of reset_noise(): drawn INSIDE the combine launch (Philox counter from the host, or from the device
        while a StepChunk is being captured) and written to weight_epsilon / bias_epsilon there; copied through from a
        staged tensor (per-update hipGraph); or, in parity mode, produced by reset_noise() from the raw draws.
        -> (fields, (weight_eps, bias_eps) tensors the backward reads)."""
        through = dict(w_eps_copy=self.weight_epsilon, b_eps_copy=self.bias_epsilon)
        if self.dev_counters is not None:
            return dict(draw=True, seed=self.seed, counter_dev=next(self.dev_counters), **through), \
                (self.weight_epsilon, self.bias_epsilon)
        if self.staged is not None and self._staged_k < len(self.staged):
            w_eps, b_eps = self.staged[self._staged_k]
            self._staged_k += 1
            return dict(w_eps=w_eps, b_eps=b_eps, **through), (w_eps, b_eps)
        if self.raw_noise is not None:
            self.reset_noise()
            return dict(w_eps=self.weight_epsilon, b_eps=self.bias_epsilon), (self.weight_epsilon, self.bias_epsilon)
        NoisyLinear._counter += 1
        return dict(draw=True, seed=self.seed, counter=NoisyLinear._counter, **through), \
            (self.weight_epsilon, self.bias_epsilon)

    def forward(self, x):
        if self.training:
            self.reset_noise()            # new noise on every training-mode forward (:90-91)
            weight = self.weight_mu + self.weight_sigma.mul(self.weight_epsilon)
            bias = self.bias_mu + self.bias_sigma.mul(self.bias_epsilon)
        else:
            weight, bias = self.weight_mu, self.bias_mu
        return small_linear(x, weight, bias)


FUSED_TREE_UPDATE = os.environ.get("GYMRL_FUSED_TREE_UPDATE", "1") != "0"  # update_priorities + the next store's priority_max in two launches (gymrl_per_update_td); False: five
OVERLAP_TREE = True     # sum-tree updates on a side stream beside the forward / backward passes (tools/micro_offpolicy.py A/B)


class _NoisyDuelingHead(torch.autograd.Function):
    """advantage / value NoisyLinear streams + the dueling combination (:108-113) on the fused layer kernels
    (csrc/lin.hip): forward = one launch building both layers' effective parameters (stacked [A + 1, K]) + one Linear
    launch whose epilogue forms q = v + a - mean(a); backward = dq -> stacked dS, input gradient, weight gradient, and
    one launch sending d mu = dW, d sigma = dW * eps of both layers into the flat gradient buffer.  The reference's
    op-by-op version of the same is ~35 launches per training-mode forward + backward."""

    @staticmethod
    def forward(ctx, x, adv, val, greedy_out, *params):
        training = adv.training
        a_wmu, a_wsig, a_bmu, a_bsig, v_wmu, v_wsig, v_bmu, v_bsig = params
        layers = [dict(w_mu=a_wmu, w_sigma=a_wsig, b_mu=a_bmu, b_sigma=a_bsig),
                  dict(w_mu=v_wmu, w_sigma=v_wsig, b_mu=v_bmu, b_sigma=v_bsig)]
        eps = []
        if training:
            for L, m in zip(layers, (adv, val)):
                fields, saved = m.noise_source()
                L.update(fields)
                eps += list(saved)
        W, b = ops.noisy_combine(layers, training)
        x = x.contiguous()
        q = ops.lin_fwd(x, W, b, ops.LIN_ACT["dueling"], argmax=greedy_out)      # greedy_out i32[B]: argmax(q) on the way
        ctx.training = training
        ctx.save_for_backward(x, W, *eps)
        ctx.sinks = [getattr(p, "_gymrl_sink", None) for p in params]
        ctx.shapes = [tuple(p.shape) for p in params]
        return q

    @staticmethod
    def backward(ctx, dq):
        x, W, *eps = ctx.saved_tensors
        dS = ops.dueling_bwd(dq.contiguous())
        dx = ops.lin_bwd_input(dS, None, W)[0] if ctx.needs_input_grad[0] else None
        grads = [None] * 8
        if any(ctx.needs_input_grad[4:]):
            dW, db = torch.empty_like(W), torch.empty(W.shape[0], dtype=W.dtype, device=W.device)
            ops.lin_bwd_weight(dS, None, x, dW, db)
            # destinations: the flat gradient buffer's views when every parameter's GradSink is armed, fresh tensors otherwise
            slots = None
            if all(s is not None and s[0].armed for s in ctx.sinks):
                slots = [s[0].direct(s[1]) for s in ctx.sinks]
                if len({a for _, a in slots}) != 1:
                    for s in ctx.sinks:
                        s[0].undo(s[1])
                    slots = None
            if slots is None:
                grads = [torch.empty(shape, dtype=W.dtype, device=W.device) for shape in ctx.shapes]
                dst, acc = grads, False
            else:
                dst, acc = [v for v, _ in slots], slots[0][1]
            layers = []
            for i in range(2):
                L = dict(w_mu=dst[4 * i], b_mu=dst[4 * i + 2],                 # (only their shapes are read by the split)
                         dw_mu=dst[4 * i], dw_sigma=dst[4 * i + 1], db_mu=dst[4 * i + 2], db_sigma=dst[4 * i + 3])
                if ctx.training:
                    L.update(w_sigma=dst[4 * i + 1], b_sigma=dst[4 * i + 3], w_eps=eps[2 * i], b_eps=eps[2 * i + 1])
                layers.append(L)
            ops.noisy_split(layers, dW, db, ctx.training, accumulate=acc)
        return (dx, None, None, None, *grads)


class DuelingNoisyNetwork(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim=256, seed=0):
        super().__init__()
        self.fc1 = SmallLinear(state_dim, hidden_dim, act="relu")
        self.fc2 = SmallLinear(hidden_dim, hidden_dim, act="relu")
        self.advantage = NoisyLinear(hidden_dim, action_dim, seed=seed)
        self.value = NoisyLinear(hidden_dim, 1, seed=seed + 1)

    def forward(self, x, greedy_out=None):
        """greedy_out: optional i32[B] that receives argmax(q) (select_action's greedy action) from the head's launch."""
        x = self.fc2(self.fc1(x))
        a, v = self.advantage, self.value
        if gnn.FUSED_LINEAR and x.is_cuda and a.out_features + 1 <= 16:
            return _NoisyDuelingHead.apply(x, a, v, greedy_out, a.weight_mu, a.weight_sigma, a.bias_mu, a.bias_sigma,
                                           v.weight_mu, v.weight_sigma, v.bias_mu, v.bias_sigma)
        advantage, value = self.advantage(x), self.value(x)
        q = value + (advantage - advantage.mean(dim=-1, keepdim=True))
        if greedy_out is not None:
            greedy_out.copy_(q.argmax(dim=-1))
        return q


class SumTree:
    """rainbow_dqn_cartpole.py:116-152 on a device float64 array."""

    def __init__(self, capacity, device):
        self.capacity, self.tree_capacity = int(capacity), 2 * int(capacity) - 1
        self.tree = torch.zeros(self.tree_capacity, dtype=torch.float64, device=device)
        self._ws = ops.per_workspace(max(8192, capacity), device)
        self._max = torch.zeros(1, dtype=torch.float64, device=device)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=device)
        self._max_fresh = False        # _max holds the maximum over the leaves as they are (update_td() just computed it)

    def update(self, data_index, priority):
        idx = torch.as_tensor([int(data_index)], dtype=torch.int32, device=self.tree.device)
        pr = torch.as_tensor([float(priority)], dtype=torch.float64, device=self.tree.device)
        self._max_fresh = False
        ops.per_update(self.tree, self.capacity, 1, self._ws, idx=idx, prio=pr)

    def update_batch(self, idx, prio):
        self._max_fresh = False
        ops.per_update(self.tree, self.capacity, idx.numel(), self._ws, idx=idx, prio=prio)

    def update_td(self, idx, td, alpha, eps):
        """update_priorities straight from the TD errors, the leaves' new maximum riding in the same two launches (the next
        store needs it: priority_max then costs nothing).  False: the batch / capacity is outside that kernel's range."""
        if idx.numel() > ops.PER_TD_MAX_BATCH or self.capacity >= 1 << 30:
            return False
        ops.per_update_td(self.tree, self.capacity, idx, td, alpha, eps, self._ws, max_out=self._max, ticket=self._ticket)
        self._max_fresh = True
        return True

    def update_range(self, start, n, priority=None, priority_dev=None, start_dev=None):
        self._max_fresh = False
        ops.per_update(self.tree, self.capacity, n, self._ws, idx_start=start, prio_scalar=priority or 0.0,
                       prio_scalar_dev=priority_dev, idx_start_dev=start_dev)

    @property
    def priority_sum(self):
        return self.tree[0]

    @property
    def priority_max(self):
        if self._max_fresh:            # (same stream order as the launches that wrote it)
            return self._max
        return ops.per_max_leaf(self.tree, self.capacity, self._max, self._ws)


class PrioritizedNStepBuffer:
    """rainbow_dqn_cartpole.py:155-264 for N env streams (one n-step window per env)."""

    def __init__(self, config, state_dim, num_envs=1, device=None, seed=0):
        self.device = torch.device(device or config.device)
        self.capacity, self.batch_size = int(config.memory_capacity), int(config.batch_size)
        self.n_steps, self.gamma, self.alpha = int(config.n_steps), float(config.gamma), float(config.alpha)
        self.beta = self.beta_init = float(config.beta_init)
        self.N, d = int(num_envs), self.device
        if self.capacity < self.N:
            raise ValueError("memory_capacity must be >= num_envs")
        self.sum_tree = SumTree(self.capacity, d)
        n, N, D = self.n_steps, self.N, state_dim
        self.win = (torch.zeros(n, N, D, device=d), torch.zeros(n, N, dtype=torch.int32, device=d),
                    torch.zeros(n, N, device=d), torch.zeros(n, N, D, device=d),
                    torch.zeros(n, N, dtype=torch.uint8, device=d), torch.zeros(n, N, dtype=torch.uint8, device=d))
        self.ring = (torch.zeros(self.capacity, D, device=d), torch.zeros(self.capacity, 1, dtype=torch.int32, device=d),
                     torch.zeros(self.capacity, device=d), torch.zeros(self.capacity, D, device=d),
                     torch.zeros(self.capacity, dtype=torch.uint8, device=d))
        self.current_size, self.count, self.pushes = 0, 0, 0
        self.seed, self.draws = seed, 0
        self._tree_ahead = None        # side stream on which stage_tree() wrote the next store's priorities

    def _new_rows_priorities(self, dev=None):
        if dev is not None:                       # StepChunk capture: the cursor of each replay lives on the device
            self.sum_tree.update_range(0, self.N, priority_dev=self.sum_tree.priority_max, start_dev=dev)
        elif self.current_size == 0:
            self.sum_tree.update_range(self.count, self.N, priority=1.0)
        else:
            self.sum_tree.update_range(self.count, self.N, priority_dev=self.sum_tree.priority_max)

    def stage_tree(self, stream, dev=None, chain=False):
        """The sum-tree half of the NEXT store_transition() — priority_max and the N new leaves with their ancestors,
        which depend on the ring cursor but not on the transitions themselves — issued now on `stream`, so that it
        runs beside the acting forward + env step + n-step push of the same vector step (round 4: the N-row store is one
        pairwise-summed addition per ancestor, ~16 us for its two launches at N = 8192 — it was a 72-79 us dependent chain
        when the overlap was built)."""
        if self._tree_ahead is not None or (dev is None and self.pushes + 1 < self.n_steps) or not OVERLAP_TREE:
            return
        if not chain:      # chain: the tree's previous writer (update_priorities) ran on `stream` itself, after the last reader
            stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self._new_rows_priorities(None if dev is None else dev[8:16])
        self._tree_ahead = stream

    def store_transition(self, state, action, reward, next_state, terminal, done, dev=None, ep_len=None, max_len=0):
        """:179-205 for N rows [N, ...]; new rows get priority 1.0 (empty buffer) or priority_max.
        dev (StepChunk capture): device record {pushes, cursor} of the replayed step; the host cursors are advanced by
        the trainer's staging loop instead, and the window is known to be full."""
        if dev is not None:
            ops.nstep_push(self.win, self.n_steps, self.n_steps, self.gamma, state, action, reward, next_state,
                           terminal, done, self.ring, 0, dev=dev, ep_len=ep_len, max_episode_steps=max_len)
            return self.after_push(True, dev)
        if not (torch.is_tensor(reward) and reward.is_cuda):
            # host scalars / numpy rows — the reference's `store_transition(state, action, reward, next_state, terminal, done)` (:380)
            N, d_ = self.N, self.ring[0].device
            D = self.ring[0].shape[1]
            state, next_state = (scalar.rows(x, N, torch.float32, d_, D) for x in (state, next_state))
            action, reward = scalar.rows(action, N, torch.int32, d_), scalar.rows(reward, N, torch.float32, d_)
            terminal, done = scalar.rows(terminal, N, torch.uint8, d_), scalar.rows(done, N, torch.uint8, d_)
        emitted = ops.nstep_push(self.win, self.n_steps, self.pushes, self.gamma, state, action, reward, next_state,
                                 terminal, done, self.ring, self.count)
        self.after_push(emitted)

    def after_push(self, emitted, dev=None):
        """The sum-tree half of store_transition() once the n-step push (gymrl_nstep_push, or the fused acting launch)
        is queued: the new rows' priorities — or the join with stage_tree()'s side stream — and the host cursors."""
        if dev is not None:
            if self._tree_ahead is not None:
                torch.cuda.current_stream().wait_stream(self._tree_ahead)
                self._tree_ahead = None
            else:
                self._new_rows_priorities(dev[8:16])
            return
        self.pushes += 1
        if emitted:
            if self._tree_ahead is not None:
                torch.cuda.current_stream().wait_stream(self._tree_ahead)
                self._tree_ahead = None
            else:
                self._new_rows_priorities()
            self.count = (self.count + self.N) % self.capacity
            self.current_size = min(self.current_size + self.N, self.capacity)
        elif self._tree_ahead is not None:
            raise RuntimeError("stage_tree() ran ahead of a push that emitted nothing")

    def draw(self, total_steps, max_train_steps, u=None, out=None, dev=None):
        """The proportional draw of :220-243 -> (batch_index i32[B], is_weight f32[B]); `out` = fixed (idx, prio, w).
        dev (StepChunk capture): device record {counter, size, beta} of the replayed step."""
        if dev is not None:
            idx, _, w = ops.per_sample(self.sum_tree.tree, self.capacity, self.batch_size, 1, 0.0, self.sum_tree._ws,
                                       seed=self.seed, out=out, dev=dev)
            return idx, w
        self.beta = self.beta_init + (1 - self.beta_init) * (total_steps / max_train_steps)
        self.draws += 1
        idx, _, w = ops.per_sample(self.sum_tree.tree, self.capacity, self.batch_size, self.current_size, self.beta,
                                   self.sum_tree._ws, u=u, seed=self.seed, counter=self.draws, out=out)
        return idx, w

    def gather(self, idx):
        s, a, r, s2, f = ops.replay_gather(self.ring, idx)
        return {"state": s, "action": a.long(), "reward": r, "next_state": s2, "terminal": f}

    def sample(self, total_steps, max_train_steps, u=None):
        """:220-256 -> (batch dict, batch_index i32[B], is_weight f32[B])."""
        idx, w = self.draw(total_steps, max_train_steps, u=u)
        return self.gather(idx), idx, w

    def update_priorities(self, batch_index, td_errors):
        """:258-261: p = (|td| + 0.01)^alpha, applied in batch order."""
        if FUSED_TREE_UPDATE and td_errors.dtype == torch.float32 and batch_index.dtype == torch.int32 \
                and self.sum_tree.update_td(batch_index, td_errors, self.alpha, 0.01):
            return
        pr = ops.per_priorities(td_errors, self.alpha, 0.01)
        self.sum_tree.update_batch(batch_index, pr)

    def __len__(self):
        return self.current_size

    def state_dict(self):
        """Ring, open n-step windows, the float64 sum tree and every cursor (PER tree state)."""
        return {"ring": [t.detach().cpu() for t in self.ring], "win": [t.detach().cpu() for t in self.win],
                "tree": self.sum_tree.tree.detach().cpu(), "current_size": self.current_size, "count": self.count,
                "pushes": self.pushes, "draws": self.draws, "beta": self.beta}

    def load_state_dict(self, sd):
        for dst, src in zip(self.ring + self.win, list(sd["ring"]) + list(sd["win"])):
            dst.copy_(src.to(dst.device))
        self.sum_tree.tree.copy_(sd["tree"].to(self.device))
        self.sum_tree._max_fresh = False
        self.current_size, self.count, self.pushes = int(sd["current_size"]), int(sd["count"]), int(sd["pushes"])
        self.draws, self.beta = int(sd["draws"]), float(sd["beta"])


class RainbowDQNTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd.RainbowDQNTrainer needs an MI355X and libgymrl_hip.so; no CPU fallback")
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        self.env = VecEnv(config.env_name, config.num_envs, device=self.device, seed=self.base_seed)
        self.state_dim, self.action_dim = self.env.observation_space.shape[0], self.env.action_space.n
        self.max_steps_per_episode = self.env.spec.max_episode_steps
        self.max_train_steps = self.max_steps_per_episode * config.max_episodes
        g = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.policy_net = DuelingNoisyNetwork(self.state_dim, self.action_dim, config.hidden_dim, seed=self.base_seed)
        torch.random.set_rng_state(g)
        self.target_net = copy.deepcopy(self.policy_net)
        self.flat_params, self.flat_grads = flatten_module(self.policy_net, self.device)
        self.target_flat, _ = flatten_module(self.target_net, self.device)
        self.target_net.eval()
        self._side = torch.cuda.Stream(device=self.device)
        self._sink = GradSink(self.policy_net)
        self.optimizer = FusedAdam(self.flat_params, self.flat_grads, lr=config.lr, eps=1e-8,
                                   max_grad_norm=config.grad_clip)
        self.memory = PrioritizedNStepBuffer(config, self.state_dim, config.num_envs, self.device, seed=self.base_seed)
        self.total_steps = 0
        self.episode_rewards = deque(maxlen=100)
        self._loss = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._parity_u = None          # tests: iterator of f64[B] PER uniforms for update()
        self._graph = None             # hipGraph of the update, captured on first use (update_async)

    # ------------------------------------------------------------ fused vector step (csrc/offpolicy_step.hip) --
    def _fused_update_ok(self):
        cfg = self.cfg
        return (bool(getattr(cfg, "fused_step", True)) and gnn.FUSED_LINEAR
                and ops.rainbow_fused_shape_ok(cfg.batch_size, self.state_dim, self.action_dim, cfg.hidden_dim))

    def _fused_act_ok(self):
        env = self.env
        return (self._fused_update_ok() and isinstance(env, VecEnv) and env.kind == ops.CARTPOLE and self.action_dim == 2
                and self.memory.capacity >= env.n)

    def _fused_state(self):
        f = getattr(self, "_fused", None)
        if f is None or f["env"] is not self.env:
            cfg, m, p = self.cfg, self.memory, self.policy_net
            A1, H, d = self.action_dim + 1, cfg.hidden_dim, self.device
            f = self._fused = dict(env=self.env, dW=torch.empty(A1, H, device=d), db=torch.empty(A1, device=d),
                                   ws=ops.rainbow_update_workspace(cfg.batch_size, self.state_dim, self.action_dim, H, d))
            f["upd"] = ops.rainbow_update_args(cfg.batch_size, self.state_dim, self.action_dim, p, self.target_net, m.ring,
                                               cfg.gamma ** cfg.n_steps, self._loss, f["dW"], f["db"], f["ws"])
            f["act"] = (ops.rainbow_act_args(self.env, p, m.win, m.ring, m.capacity, m.n_steps, m.gamma, self.max_steps_per_episode)
                        if isinstance(self.env, VecEnv) else None)
            # MFMA-operand images of fc2 (policy forward / input-gradient, target forward; csrc/lin_device.hpp): never kept
            # across parameter writes — every consumer's own gymrl_noisy_combine launch rebuilds the ones it reads
            f["img"] = (torch.empty(3, H * H, device=d) if H % 16 == 0 and getattr(cfg, "fused_images", True) else None)
        return f

    def _fc2_images(self, update=True):
        """noisy_combine(images=...) entries: policy fc2 forward (+ input-gradient and the target's forward for the update)."""
        img = self._fused_state()["img"]
        if img is None:
            return None
        if not update:
            return [(self.policy_net.fc2.weight, img[0], None)]
        return [(self.policy_net.fc2.weight, img[0], img[1]), (self.target_net.fc2.weight, img[2], None)]

    @staticmethod
    def _noisy_fields(m, **kw):
        return dict(w_mu=m.weight_mu, w_sigma=m.weight_sigma, b_mu=m.bias_mu, b_sigma=m.bias_sigma, **kw)

    @torch.no_grad()
    def select_action(self, state, deterministic=False, count=True):
        """:293-309 for a batch [N, D]: greedy on the noisy Q (no epsilon).  count=False: the caller advances
        total_steps itself (StepChunk staging)."""
        state, kind = scalar.obs_batch(state, self.device)       # ONE host observation in -> python int out (:293-309)
        if not deterministic and count:
            self.total_steps += state.shape[0]
        if deterministic:
            self.policy_net.eval()
        action = torch.empty(state.shape[0], dtype=torch.int32, device=state.device)
        self.policy_net(state, greedy_out=action)
        if deterministic:
            self.policy_net.train()
        return scalar.discrete_out(action, kind)

    def update(self, u=None):
        """:311-361.  Returns the loss as a python float."""
        cfg = self.cfg
        if len(self.memory) < cfg.batch_size:
            return 0.0
        if u is None and self._parity_u is not None:
            u = next(self._parity_u)
        batch_index, is_weight = self.memory.draw(self.total_steps, self.max_train_steps, u=u)
        self._update_body(batch_index, is_weight)
        self._anneal_lr()
        return float(self._loss.item()) / cfg.batch_size

    def _update_body(self, batch_index, is_weight, bias=None, join=True):
        """Everything after the proportional draw; bias = f32[4] device view of Adam's step scalars under a hipGraph.
        join=False (inside a StepChunk): the side stream keeps the sum tree — it goes straight on to the next vector
        step's new rows (stage_tree(chain=True)) and is joined when that step stores its transitions."""
        cfg = self.cfg
        if self._fused_update_ok() and batch_index.numel() == cfg.batch_size:
            return self._update_body_fused(batch_index, is_weight, bias, join)
        state, action, reward, next_state, terminal = ops.replay_gather(self.memory.ring, batch_index)
        fused = gnn.FUSED_LINEAR and self.policy_net.advantage.out_features + 1 <= 16
        if fused:
            q_next_online, q_next_target, q, saved = self._three_forwards(next_state, state)
        else:
            with torch.no_grad():
                q_next_online = self.policy_net(next_state)               # fresh noise (:320)
                q_next_target = self.target_net(next_state)               # eval mode: mu weights only
            q = self.policy_net(state)                                    # fresh noise again (:334)
        self._loss.zero_()
        td, dq = ops.dqn_td_loss(q, q_next_target, action.view(-1), reward, terminal, cfg.gamma ** cfg.n_steps,
                                 q_next_online=q_next_online, w=is_weight, loss_sum=self._loss)
        # :340 update_priorities before backward — nothing below reads the tree, so it runs on a side stream (a fork
        # inside the captured graph) beside the backward pass, Adam and the Polyak update, and joins at the end
        # (the main path's launches are issued first: the graph executor keeps the first-recorded branch on the
        # queue it was on and wakes another queue for the second, which costs that branch 30-60 us)
        main, side = torch.cuda.current_stream(), self._side if OVERLAP_TREE else torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        if fused:
            self._backward(dq, saved)
        else:
            self._sink.arm()
            q.backward(dq)
            self._sink.collect()
        self.optimizer.step(bias_dev=bias, polyak=(self.target_flat, cfg.tau))   # clip_grad_norm_(10) + Adam, then the soft
        #                                                                    target update :347-352 (parameters only) in the launch
        with torch.cuda.stream(side):
            side.wait_event(fork)
            self.memory.update_priorities(batch_index, td)
        if join:
            main.wait_stream(side)
        else:
            self._tree_keep = (td, batch_index)       # read on the side stream: alive until the next join

    @torch.no_grad()
    def _noisy_heads3(self, with_acting=False, images=False):
        """The stacked effective head parameters of the update's three passes — policy on s' (first draw :320), target on s'
        (means), policy on s (second draw :334) — built by ONE gymrl_noisy_combine launch, both draws made inside it in the
        eager order.  -> (W [3 (A + 1), H], b [3 (A + 1)], the second draw's epsilons for the backward).  with_acting: the
        vector step's acting forward (its own draw, made FIRST: the eager order) rides in the same launch, its rows in front.
        images: fc2's weight images are rebuilt on extra workgroups of the launch (the caller consumes them before the next
        parameter write)."""
        p, t = self.policy_net, self.target_net
        acting = []
        if with_acting:       # (no *_copy: the module's epsilon buffers end the step holding the second draw, as in the eager order)
            acting = [self._noisy_fields(m, **{k: v for k, v in m.noise_source()[0].items() if not k.endswith("_copy")})
                      for m in (p.advantage, p.value)]
        first = []
        for m in (p.advantage, p.value):
            f = {k: v for k, v in m.noise_source()[0].items() if not k.endswith("_copy")}   # not needed after the launch
            if f.get("w_eps") is m.weight_epsilon:        # parity mode: raw draws land in the module's buffers, which the
                f["w_eps"], f["b_eps"] = f["w_eps"].clone(), f["b_eps"].clone()      # second draw below overwrites
            first.append(self._noisy_fields(m, **f))
        target = [self._noisy_fields(m, eval=True) for m in (t.advantage, t.value)]
        second, eps = [], []
        for m in (p.advantage, p.value):
            f, saved = m.noise_source()
            second.append(self._noisy_fields(m, **f))
            eps += list(saved)
        W, b = ops.noisy_combine(acting + first + target + second, training=True, images=self._fc2_images() if images else None)
        return W, b, eps

    @torch.no_grad()
    def _update_body_fused(self, batch_index, is_weight, bias=None, join=True, heads=None, images=None):
        """_update_body with everything between the proportional draw and the optimiser step as gymrl_rainbow_update's two
        launches (+ the two NoisyLinear launches that own the noise bookkeeping): same values, same destinations."""
        cfg, p = self.cfg, self.policy_net
        f = self._fused_state()
        if heads is None:            # the launch that builds the heads also rebuilds fc2's weight images for the row launch
            W, b, eps = self._noisy_heads3(images=True)
            images = f["img"]
        else:                        # the caller's launch did (images) or did not (None: fc2 is read in place)
            W, b, eps = heads
        td = torch.empty(cfg.batch_size, device=self.device)
        # the stacked head's gradient is split into d mu / d sigma of the two NoisyLinear layers by the weight-gradient launch
        split = [(m.weight_mu.grad, m.weight_sigma.grad, m.bias_mu.grad, m.bias_sigma.grad, eps[2 * i], eps[2 * i + 1])
                 for i, m in enumerate((p.advantage, p.value))]
        ops.rainbow_update(f["upd"], batch_index, is_weight, W, b, td, split=split, phase=1, images=images)   # rows: td is complete
        # :340 update_priorities needs only td: it forks off here, beside the weight gradients, the clip and Adam (the sum
        # tree's chain — priorities, then the next step's new rows, then the draw — is the step's critical path)
        main, side = torch.cuda.current_stream(), self._side if OVERLAP_TREE else torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        ops.rainbow_update(f["upd"], batch_index, is_weight, W, b, td, split=split, phase=2, images=images)   # tiles
        self.optimizer.step(bias_dev=bias, polyak=(self.target_flat, cfg.tau))
        with torch.cuda.stream(side):
            side.wait_event(fork)
            self.memory.update_priorities(batch_index, td)
        if join:
            main.wait_stream(side)
        else:
            self._tree_keep = (td, batch_index)       # read on the side stream: alive until the next join

    @torch.no_grad()
    def _three_forwards(self, next_state, state):
        """policy_net(next_state) [fresh noise, :320], target_net(next_state) [eval: mu only] and policy_net(state)
        [fresh noise again, :334] layer by layer, the three passes sharing every launch (csrc/lin.hip items): 2 trunk
        launches, one launch building the six NoisyLinear layers' effective parameters (both training-mode draws made
        inside it, in the eager order), one head launch with the dueling epilogue.  -> q_next_online, q_next_target, q
        and what _backward() needs of the third pass.  Same arithmetic as the module path, kernel for kernel."""
        p, t = self.policy_net, self.target_net
        relu, duel = ops.LIN_ACT["relu"], ops.LIN_ACT["dueling"]
        h1 = ops.lin_fwd([next_state, next_state, state], [p.fc1.weight, t.fc1.weight, p.fc1.weight],
                         [p.fc1.bias, t.fc1.bias, p.fc1.bias], relu)
        h2 = ops.lin_fwd(h1, [p.fc2.weight, t.fc2.weight, p.fc2.weight], [p.fc2.bias, t.fc2.bias, p.fc2.bias], relu)

        def fields(m, **kw):
            return dict(w_mu=m.weight_mu, w_sigma=m.weight_sigma, b_mu=m.bias_mu, b_sigma=m.bias_sigma, **kw)
        first = []
        for m in (p.advantage, p.value):
            f = {k: v for k, v in m.noise_source()[0].items() if not k.endswith("_copy")}   # not needed after the launch
            if f.get("w_eps") is m.weight_epsilon:        # parity mode: raw draws land in the module's buffers, which the
                f["w_eps"], f["b_eps"] = f["w_eps"].clone(), f["b_eps"].clone()      # second draw below overwrites
            first.append(fields(m, **f))
        target = [fields(m, eval=True) for m in (t.advantage, t.value)]
        second, eps = [], []
        for m in (p.advantage, p.value):
            f, saved = m.noise_source()
            second.append(fields(m, **f))
            eps += list(saved)
        W, b = ops.noisy_combine(first + target + second, training=True)
        A1 = p.advantage.out_features + 1
        Ws, bs = [W[i * A1:(i + 1) * A1] for i in range(3)], [b[i * A1:(i + 1) * A1] for i in range(3)]
        q_no, q_nt, q = ops.lin_fwd(h2, Ws, bs, duel)
        return q_no, q_nt, q, (state, h1[2], h2[2], Ws[2], eps)

    @torch.no_grad()
    def _backward(self, dq, saved):
        """loss.backward() (:342) of the third pass, by hand on the layer kernels: every gradient lands in its view of
        the flat gradient buffer (overwritten: the optimiser step left it zeroed or stale)."""
        state, h1, h2, W, eps = saved
        p = self.policy_net
        relu = ops.LIN_ACT["relu"]
        dS = ops.dueling_bwd(dq)
        dh2 = ops.lin_bwd_input(dS, None, W)[0]
        dW, db = torch.empty_like(W), torch.empty(W.shape[0], dtype=W.dtype, device=W.device)
        ops.lin_bwd_weight(dS, None, h2, dW, db)
        layers = []
        for i, m in enumerate((p.advantage, p.value)):
            layers.append(dict(w_mu=m.weight_mu.grad, b_mu=m.bias_mu.grad, w_sigma=m.weight_sigma.grad,
                               b_sigma=m.bias_sigma.grad, w_eps=eps[2 * i], b_eps=eps[2 * i + 1],
                               dw_mu=m.weight_mu.grad, dw_sigma=m.weight_sigma.grad, db_mu=m.bias_mu.grad,
                               db_sigma=m.bias_sigma.grad))
        ops.noisy_split(layers, dW, db, training=True)
        dh1 = ops.lin_bwd_input(dh2, h2, p.fc2.weight, relu)[0]
        ops.lin_bwd_weight(dh2, h2, h1, p.fc2.weight.grad, p.fc2.bias.grad, relu)
        ops.lin_bwd_weight(dh1, h1, state, p.fc1.weight.grad, p.fc1.bias.grad, relu)

    def _anneal_lr(self):
        cfg = self.cfg
        lr_now = 0.9 * cfg.lr * (1 - self.total_steps / self.max_train_steps) + 0.1 * cfg.lr     # :354-356
        for param_group in self.optimizer.param_groups:
            param_group["lr"] = lr_now

    def _draw_buffers(self):
        """Fixed (index, priority, weight) buffers of the proportional draw: captured graphs hold their addresses."""
        if getattr(self, "_g_draw", None) is None:
            B, d = self.cfg.batch_size, self.device
            self._g_draw = (torch.empty(B, dtype=torch.int32, device=d), torch.empty(B, dtype=torch.float64, device=d),
                            torch.empty(B, dtype=torch.float32, device=d))
        return self._g_draw

    def update_async(self):
        """update() without the host round trip: the counter-keyed launches (proportional draw, NoisyNet noise of the
        two training-mode forwards) and one scalar store run eagerly, then the captured hipGraph of `_update_body`."""
        cfg, m = self.cfg, self.memory
        if len(m) < cfg.batch_size:
            return
        if self._graph is None:
            from .graphs import GraphedStep, StepScalars
            self._scalars = StepScalars(self.device)
            bias, self._off = self._scalars.slot(16, torch.float32)
            self._draw_buffers()
            self._graph = GraphedStep(lambda: self._update_body(self._g_draw[0], self._g_draw[2], bias=bias))
        m.draw(self.total_steps, self.max_train_steps, out=self._g_draw)
        layers = (self.policy_net.advantage, self.policy_net.value)
        c0 = NoisyLinear._counter                            # the eager path's order: (advantage, value) per forward
        NoisyLinear._counter += 4
        for j, layer in enumerate(layers):
            layer.draw_staged_at([c0 + 1 + j, c0 + 3 + j])
        self._scalars.set(self._off, self.optimizer.next_bias())
        self._scalars.flush()
        self._graph()
        for layer in layers:
            layer._staged_k = len(layer.staged)              # consumed by the replay; later forwards draw their own
        self._anneal_lr()

    def save_checkpoint(self, path, include_memory=True):
        """ModelLoader-style dict incl. the NoisyNet draw counter and — unlike the reference, which
        skips `memory` — the prioritised n-step buffer with its sum tree."""
        from .utils import checkpoint
        extra = {"memory_state_dict": self.memory.state_dict()} if include_memory else {}
        return checkpoint.save_agent(path, {"policy_net": self.policy_net, "target_net": self.target_net},
                                     {"optimizer": (self.policy_net, self.optimizer)}, total_steps=self.total_steps,
                                     noisy_counter=NoisyLinear._counter, episode_rewards=list(self.episode_rewards), **extra)

    def load_checkpoint(self, path):
        from .utils import checkpoint
        rest = checkpoint.load_agent(path, {"policy_net": self.policy_net, "target_net": self.target_net},
                                     {"optimizer": (self.policy_net, self.optimizer)})
        self.total_steps = int(rest["total_steps"])
        NoisyLinear._counter = int(rest["noisy_counter"])
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
                                   term=torch.zeros(N, dtype=torch.uint8, device=d),
                                   ep_len=torch.zeros(N, dtype=torch.int32, device=d),
                                   term_b=torch.zeros(N, dtype=torch.bool, device=d),
                                   tracker=EpisodeTracker(N, d, flush_every=1 if N == 1 else self.CHUNK))
        lb["tracker"].k, lb["tracker"].episodes = 0, 0
        return lb

    def _vector_step(self, lb, obs, nxt, ep_ret, done, rec=None, chain=False, act_heads=None, draw_dev=None, fc2_img=None):
        """One vector step of :363-405 up to (not including) the update.  rec: this step's StepChunk record views
        when the step is being captured (every per-step scalar then comes from the device).  draw_dev: the step's
        `sample` record — the proportional draw (:220-243) reads the tree and the record, not the transitions, so it
        follows the new priorities on the tree's stream, beside the acting launch; returns whether it was issued."""
        env, m = self.env, self.memory
        push = None if rec is None else rec["push"]
        m.stage_tree(self._side, dev=push, chain=chain)   # this step's new priorities, beside the acting forward + env step
        drawn = False
        if draw_dev is not None and m._tree_ahead is not None:
            with torch.cuda.stream(m._tree_ahead):
                m.draw(0, 1, out=self._g_draw, dev=draw_dev)
            drawn = True
        if rec is not None and act_heads is None:
            for layer, c in ((self.policy_net.advantage, rec["noise"][0:8]), (self.policy_net.value, rec["noise"][8:16])):
                layer.dev_counters = iter([c])
        if self._fused_act_ok():
            # greedy acting on the noisy Q (:371), env.step, `terminal` (:376) and the n-step push as ONE launch behind the
            # launch that builds the heads' effective parameters (this forward's NoisyNet draw is made inside it)
            p = self.policy_net
            if act_heads is None:
                layers = [self._noisy_fields(mod, **mod.noise_source()[0]) for mod in (p.advantage, p.value)]
                ims = self._fc2_images(update=False)
                W, b = ops.noisy_combine(layers, training=True, images=ims)
                fc2_img = None if ims is None else ims[0][1]
            else:                     # fc2_img: rebuilt by the caller's launch, or None
                W, b = act_heads
            if rec is None:
                self.total_steps += env.n                 # select_action's count (:301)
            emitted = ops.rainbow_act_step(self._fused_state()["act"], env, obs, nxt, W, b, pushes=m.pushes, cursor=m.count,
                                           push_dev=push, done_out=done, ep_ret_out=ep_ret, ep_stats=env.ep_stats, fc2_img=fc2_img)
            m.after_push(emitted, push)
            return drawn
        action = self.select_action(obs) if rec is None else self.select_action(obs, count=False)
        env.step(action, nxt, lb["rew"], done_out=done, term_obs_out=lb["tobs"], ep_ret_out=ep_ret, ep_len_out=lb["ep_len"])
        # :376 terminal = done and step != max_steps_per_episode - 1: decided by the step INDEX inside the
        # episode, not by gymnasium's terminated flag (a pole that falls exactly on the last step of the
        # time limit is stored as non-terminal, an early truncation as terminal)
        if rec is None:
            torch.logical_and(done.bool(), lb["ep_len"] != self.max_steps_per_episode, out=lb["term_b"])
            lb["term"].copy_(lb["term_b"])
            m.store_transition(obs, action, lb["rew"], lb["tobs"], lb["term"], done)
        else:                                                        # the same flag, formed inside the n-step push
            m.store_transition(obs, action, lb["rew"], lb["tobs"], None, done, dev=push, ep_len=lb["ep_len"],
                               max_len=self.max_steps_per_episode)
        return drawn

    def _chunk_body(self, lb, j):
        """Vector step j of a StepChunk capture: acting + env + store + proportional draw + update, every per-step
        scalar read from record j."""
        ch, tr = self._chunk, lb["tracker"]
        rec = {"noise": ch.view(j, "noise"), "push": ch.view(j, "push")}
        obs, nxt = (lb["obs"], lb["nxt"]) if j % 2 == 0 else (lb["nxt"], lb["obs"])
        if j == 0:
            # `_max_fresh` is a HOST flag read while capturing: whatever ran between two replays (an eager update_batch, a
            # load_checkpoint) may have changed the leaves without this graph knowing — its first store always recomputes
            # priority_max (one per_max_leaf launch per 16 vector steps); steps 1.. follow the chunk's own update_td
            self.memory.sum_tree._max_fresh = False
        noise = ch.view(j, "noise")
        layers = (self.policy_net.advantage, self.policy_net.value)
        if self._fused_act_ok() and self._fused_update_ok():
            # ONE NoisyLinear launch per vector step: the parameters do not change between the acting forward and the update's
            # three passes, so the acting heads (draw 1) and the update's (draws 2 and 3, the target's means) are built together —
            # eight layer entries, the draws in the eager order — and a launch leaves the chain adam -> heads -> acting
            for i, layer in enumerate(layers):
                layer.dev_counters = iter([noise[8 * i:8 + 8 * i], noise[16 + 8 * i:24 + 8 * i], noise[32 + 8 * i:40 + 8 * i]])
            W, b, eps = self._noisy_heads3(with_acting=True, images=True)
            img = self._fused_state()["img"]
            A1 = self.policy_net.advantage.out_features + 1
            drawn = self._vector_step(lb, obs, nxt, tr.ret[j], tr.done[j], rec, chain=j > 0 and OVERLAP_TREE,
                                      act_heads=(W[:A1], b[:A1]), draw_dev=ch.view(j, "sample"),
                                      fc2_img=None if img is None else img[0])
            if not drawn:
                self.memory.draw(0, 1, out=self._g_draw, dev=ch.view(j, "sample"))
            self._update_body_fused(self._g_draw[0], self._g_draw[2], bias=ch.view(j, "adam", torch.float32),
                                    join=j == ch.K - 1 or not OVERLAP_TREE, heads=(W[A1:], b[A1:], eps), images=img)
        else:
            self._vector_step(lb, obs, nxt, tr.ret[j], tr.done[j], rec, chain=j > 0 and OVERLAP_TREE)
            self.memory.draw(0, 1, out=self._g_draw, dev=ch.view(j, "sample"))
            for i, layer in enumerate(layers):             # the eager order: (advantage, value) per training-mode forward
                layer.dev_counters = iter([noise[16 + 8 * i:24 + 8 * i], noise[32 + 8 * i:40 + 8 * i]])
            self._update_body(self._g_draw[0], self._g_draw[2], bias=ch.view(j, "adam", torch.float32),
                              join=j == ch.K - 1 or not OVERLAP_TREE)
        for layer in layers:
            layer.dev_counters = None

    def _stage_chunk(self):
        """The host's bookkeeping of the next CHUNK vector steps, in the eager loop's order, written into the records."""
        ch, m, N = self._chunk, self.memory, self.env.n
        for j in range(ch.K):
            c = NoisyLinear._counter                     # acting forward: +1, +2; the update's two forwards: +3 .. +6
            ch.set(j, "noise", *(c + 1 + i for i in range(6)))
            NoisyLinear._counter += 6
            self.total_steps += N                        # select_action
            ch.set(j, "push", m.pushes, m.count)         # store_transition
            m.pushes += 1
            m.count = (m.count + N) % m.capacity
            m.current_size = min(m.current_size + N, m.capacity)
            m.beta = m.beta_init + (1 - m.beta_init) * (self.total_steps / self.max_train_steps)      # draw
            m.draws += 1
            ch.set(j, "sample", m.draws, m.current_size, m.beta)
            ch.set_bytes(j, "adam", self.optimizer.next_bias())      # update: Adam with the current rate, then the anneal
            self._anneal_lr()
        ch.flush()

    def _train(self, max_vector_steps=None):
        """:363-405 with N lock-stepped envs.  With hipGraphs on, CHUNK whole vector steps (acting, env step, n-step
        store, sum-tree updates, proportional draw, update) replay as one graph (gymrl_amd/graphs.py StepChunk)."""
        cfg, env = self.cfg, self.env
        N, D = env.n, env.obs_dim
        lb = self._loop_buffers(N, D)
        obs, nxt, tracker = lb["obs"], lb["nxt"], lb["tracker"]
        env.reset(obs)
        step = 0
        pending = None            # drain_async() token of the last chunk, collected one chunk later
        graphed = (bool(getattr(cfg, "use_graphs", True)) and self._parity_u is None
                   and self.policy_net.advantage.raw_noise is None)
        chunked = graphed and N > 1 and cfg.updates_per_step == 1 and getattr(cfg, "chunk_steps", self.CHUNK) > 0 and (self.memory.capacity & (self.memory.capacity - 1)) == 0 \
            and self.memory.capacity % N == 0
        limit = max_vector_steps or (cfg.max_episodes * cfg.max_steps // N + 1)
        while tracker.episodes < cfg.max_episodes and step < limit:
            m = self.memory
            if (chunked and tracker.k == 0 and limit - step >= self.CHUNK and obs is lb["obs"] and m.pushes + 1 >= m.n_steps
                    and m.current_size >= max(cfg.batch_size, 1) and m.count % N == 0):
                if getattr(self, "_chunk", None) is None:
                    from .graphs import StepChunk
                    self._chunk = StepChunk(self.device, self.CHUNK, [("noise", "6Q"), ("push", "2q"), ("sample", "Qqd"),
                                                                      ("adam", "4f")])
                    self._draw_buffers()
                self._stage_chunk()
                self._chunk.run(lambda j: self._chunk_body(lb, j), key=(id(env), env.state.data_ptr()))
                step += self.CHUNK
                tracker.k = self.CHUNK
                # the chunk's episode returns come back through a pinned buffer one chunk late: the host goes straight on
                # to staging the next chunk while this one runs (a sync here left the GPU idle for the host work per chunk)
                token = tracker.drain_async()
                tracker.collect(pending, self.episode_rewards)
                pending = token
                # ... except where the lag could change WHEN training stops (:398-401's `mean(last 100) >= 495`, the episode
                # budget): within reach of either rule this chunk's returns are collected at once, as the eager loop does
                if (cfg.max_episodes - tracker.episodes <= N * self.CHUNK
                        or (len(self.episode_rewards) >= 100 and np.mean(self.episode_rewards) >= 0.9 * 495.0)):
                    tracker.collect(pending, self.episode_rewards)
                    pending = None
                if len(self.episode_rewards) >= 100 and np.mean(self.episode_rewards) >= 495.0:
                    break
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
            if len(self.episode_rewards) >= 100 and np.mean(self.episode_rewards) >= 495.0:
                break
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
            env.step(act, nxt, rew, done_out=done, ep_ret_out=ep_ret)
            result = torch.where(done.bool() & torch.isnan(result), ep_ret, result)
            obs, nxt = nxt, obs
            if not torch.isnan(result).any():
                break
        return result.tolist()

    def test(self):
        return self.eval(num_episodes=5)


if __name__ == "__main__":       # python -m gymrl_amd.rainbow_dqn_cartpole [--<Config attribute> <value> ...]  (rainbow_dqn_cartpole.py:449-465)
    from .utils.cli import run_script
    run_script(Config, RainbowDQNTrainer)
