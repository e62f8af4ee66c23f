"""Recurrent PPO (mHC or PSCN backbone -> GRU -> actor / critic heads, RND intrinsic reward, sequence
minibatches with stored initial hidden states, masked means over the entropy-ratio mask, clipped value
loss) — MI355X engine behind the reference's algorithms/ppo_lstm_lunarlander.py surface:
Config :19-53, ActorCritic :56-131, PSCN :405-446, URNN :449-491, RND :494-513, RolloutBuffer :516-538,
PPOTrainer :541-851 (collect_experience :563-617, compute_advantages :619-644, masked_mean :646-655,
update_model :657-812).

What runs where: env stepping, categorical sampling (+ behaviour entropy), the RND reward, decoupled-lambda
GAE, the L4 loss forward/backward (`gymrl_ppo_rnn_loss_fwd_bwd`), the GRU cell's pointwise half
(`gymrl_gru_cell_fwd/_bwd` behind a torch.autograd.Function), clip-norm + Adam and the gradient all-reduce are
the HIP / RCCL path; the dense layers (backbone, gate GEMMs, heads, RND towers) are PyTorch-ROCm library work.
Parameter names are the reference's (`rnn.rnn.weight_ih_l0`, ...), so its state_dicts load unchanged.

Vectorisation: N envs step in lock-step; a "sequence" is `seq_len` consecutive steps of ONE env, numbered
window-major (s = window * N + env), which for N = 1 is the reference's `view(num_sequences, seq_len)`.
As in the reference, a window that crosses an episode end keeps integrating the GRU through it in the update
(the stored initial state of the NEXT window is the reset one); only collection zeroes the state on done.
"""
from collections import deque

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as gdist
from . import ops
from .graphs import capture as gcapture
from .envs import VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .ppo_full_lunarlander import MLP, PSCN, MHCBackbone, RMSNorm, cov_clip_mask  # noqa: F401  (part of this module's surface)


class Config:
    def __init__(self):
        self.env_name = "LunarLander-v3"
        self.seed = None
        self.use_mhc = True
        self.mhc_dim = 256
        self.mhc_rate = 2
        self.mhc_layers = 2
        self.mhc_sk_it = 10
        self.max_train_steps = 5e6
        self.update_freq = 4096            # steps PER ENV per rollout
        self.num_epochs = 4
        self.seq_len = 8
        self.batch_size = 128              # sequences per minibatch
        self.gamma = 0.995
        self.lam_actor = 0.95
        self.lam_critic = 0.95
        self.clip_eps_min = 0.2
        self.clip_eps_max = 0.28
        self.clip_cov_ratio = 0.0
        self.clip_cov_min = 1.0
        self.clip_cov_max = 5.0
        self.dual_clip = 3.0
        self.entropy_coef = 0.015
        self.erc_beta_low = 0.06
        self.erc_beta_high = 0.06
        self.lr = 3e-4
        self.max_grad_norm = 0.5
        self.anneal = True
        self.device = "cuda"
        # --- vectorised-engine additions ---
        self.num_envs = 1
        self.use_graphs = True             # replay the rollout forward and the minibatch body as captured hipGraphs (1 GPU)
        self.rnn_hidden = 512              # the reference hard-codes these three (:84-95)
        self.head_hidden = 512
        self.rnd_embed = 512


class _GRUCell(torch.autograd.Function):
    """h' = GRU pointwise(gi, gh, h) on the HIP kernels; the gates are recomputed in backward."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        ctx.save_for_backward(gi, gh, h)
        return ops.gru_cell_fwd(gi, gh, h)

    @staticmethod
    def backward(ctx, dh_out):
        gi, gh, h = ctx.saved_tensors
        return ops.gru_cell_bwd(gi, gh, h, dh_out.contiguous())


class URNN(nn.Module):
    """One-layer batch_first GRU with the hidden state carried as [B, H] (:449-491).  `self.rnn` is a
    torch.nn.GRU used as the parameter container (reference key names and init); its own forward is never
    called — the gate GEMMs are F.linear and the cell is `_GRUCell`."""

    def __init__(self, input_size, hidden_size, layer=nn.GRU):
        super().__init__()
        if layer is not nn.GRU:
            raise NotImplementedError("only the GRU the reference's ActorCritic instantiates (:84-88) is built")
        self.input_size, self.hidden_size, self.chunk_size = input_size, hidden_size, 1
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, batch_first=True)

    def forward(self, x, hidden_state):
        B, L = x.shape[0], x.shape[1]
        r = self.rnn
        h = torch.zeros(B, self.hidden_size, device=x.device) if hidden_state is None else hidden_state
        gi = F.linear(x.transpose(0, 1), r.weight_ih_l0, r.bias_ih_l0)       # [L, B, 3H]: one GEMM for the window
        outs = []
        for step in range(L):
            h = _GRUCell.apply(gi[step], F.linear(h, r.weight_hh_l0, r.bias_hh_l0), h)
            outs.append(h)
        return torch.stack(outs, dim=1), h


class RND(nn.Module):
    def __init__(self, input_dim, embed_dim):
        super().__init__()
        if embed_dim < 16 or embed_dim & (embed_dim - 1):
            raise ValueError("embed_dim must be a power of 2, >= 16")
        depth = int(np.log2(embed_dim // 16))
        self.predictor = PSCN(input_dim, embed_dim, depth)
        self.target = PSCN(input_dim, embed_dim, depth)
        for p in self.target.parameters():
            p.requires_grad = False

    def forward(self, x):
        with torch.no_grad():
            target = self.target(x)
        return self.predictor(x), target


class ActorCritic(nn.Module):
    def __init__(self, state_dim, action_dim, config=None):
        super().__init__()
        cfg = config or Config()
        H, Hh = getattr(cfg, "rnn_hidden", 512), getattr(cfg, "head_hidden", 512)
        if getattr(cfg, "use_mhc", True):
            self.shared = MHCBackbone(state_dim, cfg.mhc_dim, cfg.mhc_rate, cfg.mhc_layers, cfg.mhc_sk_it)
            shared_out = cfg.mhc_dim
        else:
            shared_out = getattr(cfg, "pscn_dim", 512)
            self.shared = PSCN(state_dim, shared_out, 5)
        self.rnn = URNN(shared_out, H)
        self.actor = MLP([H, Hh, action_dim], last_std=0.001)
        self.critic = MLP([H, Hh, 1], last_std=1.0)
        self.rnd = RND(state_dim, getattr(cfg, "rnd_embed", 512))

    def forward(self, x, hidden_state):
        """x [B, D] or [B, L, D] -> (logits, value, new_hidden [B, H], predict, target) (:98-116)."""
        seq = x.dim() == 3
        if not seq:
            x = x.unsqueeze(1)
        B, L = x.shape[0], x.shape[1]
        predict, target = self.rnd(x)
        feat = self.shared(x.reshape(B * L, -1)).view(B, L, -1)       # the backbone is per-token
        rnn_out, new_hidden = self.rnn(feat, hidden_state)
        logits, value = self.actor(rnn_out), self.critic(rnn_out)
        if not seq:
            logits, value, predict, target = logits.squeeze(1), value.squeeze(1), predict.squeeze(1), target.squeeze(1)
        return logits, value, new_hidden, predict, target

    @torch.no_grad()
    def get_action(self, x, hidden_state, deterministic=False, seed=0, counter=0, env_id0=0, noise_exp=None):
        """:118-131 batched -> (action i32[N], logp[N], value[N], new_hidden, predict, target, entropy[N])."""
        logits, value, new_hidden, predict, target = self.forward(x, hidden_state)
        act, logp, ent, val = ops.categorical_sample(logits, value=value.reshape(-1), noise_exp=noise_exp, seed=seed,
                                                     counter=counter, env_id0=env_id0, deterministic=deterministic)
        return act, logp, val, new_hidden, predict, target, ent

    @torch.no_grad()
    def get_value(self, x, hidden_state):
        return self.forward(x, hidden_state)[1].reshape(-1)


class RolloutBuffer:
    """[T][N] slabs incl. the pre-step hidden states and the behaviour-policy entropies (:516-538)."""

    def __init__(self, T, N, obs_dim, hidden, device):
        z = lambda *s, **k: torch.zeros(*s, device=device, **k)   # noqa: E731
        self.T, self.N = T, N
        self.states = z(T + 1, N, obs_dim)
        self.actions = z(T, N, dtype=torch.int32)
        self.log_probs, self.values, self.rewards, self.old_entropies = z(T, N), z(T, N), z(T, N), z(T, N)
        self.dones = z(T, N, dtype=torch.uint8)
        self.hidden_states = z(T, N, hidden)
        self.ep_returns = z(T, N)
        self.next_value = z(N)

    def clear(self):
        pass


class PPOTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd recurrent PPO needs an MI355X and libgymrl_hip.so; no CPU fallback")
        N, T, L = int(config.num_envs), int(config.update_freq), int(config.seq_len)
        assert T % L == 0                                                   # :544
        self.num_sequences = T // L * N
        assert self.num_sequences % int(config.batch_size) == 0             # :546
        self.rank, self.world_size = gdist.rank(), gdist.world_size()
        self.collective = gdist.collectives_active()          # world_size > 1 (or one rank under GYMRL_FORCE_COLLECTIVES)
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        self.env = VecEnv(config.env_name, N, device=self.device, seed=self.base_seed, env_id0=self.rank * N)
        state_dim, action_dim = self.env.observation_space.shape[0], self.env.action_space.n
        g = torch.random.get_rng_state()
        torch.manual_seed(self.base_seed)
        self.model = ActorCritic(state_dim, action_dim, config=config)
        torch.random.set_rng_state(g)
        self.flat_params, self.flat_grads = flatten_module(self.model, self.device)
        gdist.broadcast(self.flat_params)
        self.optimizer = FusedAdam(self.flat_params, self.flat_grads, lr=config.lr, eps=1e-5,
                                   max_grad_norm=config.max_grad_norm)
        self.hidden_size = self.model.rnn.hidden_size * self.model.rnn.chunk_size
        self.step_count = 0
        self.rollout_count = 0
        self.episode_rewards = deque(maxlen=10)
        self.lr, self.ent_coef = config.lr, config.entropy_coef
        self.buffer = RolloutBuffer(T, N, state_dim, self.hidden_size, self.device)
        self._perm_gen = torch.Generator(device=self.device)
        self._perm_gen.manual_seed(self.base_seed * 7919 + 23 + self.rank)
        self._sink = GradSink(self.model)
        self._g_seq, self._g_warm = None, 0            # hipGraph replay of the minibatch body (update_model)
        self._fwd_graph, self._fwd_in, self._fwd_out, self._fwd_warm = None, None, None, 0
        self._parity_noise = None      # tests: f32[T, N, A] Exp(1) draws of one rollout (list of them, popped per rollout)
        self._parity_perms = None      # tests: iterator of i32[num_sequences] permutations, one per epoch
        self.grad_norms = None         # tests: set to [] to record the pre-clip gradient norm of every minibatch

    @torch.no_grad()
    def collect_experience(self):
        """:563-617 for N envs."""
        b, env, cfg = self.buffer, self.env, self.cfg
        seed = (self.base_seed if cfg.seed is not None else self.base_seed + 0x9E3779B1 * (self.rollout_count + 1))
        env.reset(b.states[0], seed=seed & 0x7FFFFFFFFFFFFFFF)
        noise = self._parity_noise.pop(0) if self._parity_noise else None
        hidden = torch.zeros(b.N, self.hidden_size, device=self.device)
        c0 = self.rollout_count * b.T
        graphed = bool(getattr(cfg, "use_graphs", True)) and self._parity_noise is None
        for t in range(b.T):
            b.hidden_states[t].copy_(hidden)
            if graphed:                                   # the ~150-launch forward as one graph launch
                logits, value, hidden, predict, target = self._forward_graphed(b.states[t], hidden)
            else:
                logits, value, hidden, predict, target = self.model(b.states[t], hidden)
            ops.categorical_sample(logits, value=value.reshape(-1), noise_exp=None if noise is None else noise[t],
                                   seed=env.seed, counter=c0 + t, env_id0=env.env_id0, act_out=b.actions[t],
                                   logp_out=b.log_probs[t], ent_out=b.old_entropies[t], value_out=b.values[t])
            env.step(b.actions[t], b.states[t + 1], b.rewards[t], done_out=b.dones[t], ep_ret_out=b.ep_returns[t])
            ops.rnd_reward(predict, target, rew_inout=b.rewards[t])          # ep_returns stay extrinsic (:590-591)
            hidden = hidden * (1 - b.dones[t].to(hidden.dtype)).unsqueeze(1)  # fresh state after an episode end (:611)
        self.step_count += b.T * b.N
        self.rollout_count += 1
        b.next_value.copy_(self.model.get_value(b.states[b.T], hidden))
        self._last_hidden = hidden

    @torch.no_grad()
    def _forward_graphed(self, x, hidden):
        """model(x, hidden) for the rollout through a captured hipGraph (fixed input / output buffers; the parameters
        are views of the flat buffer the optimiser updates in place, so one capture serves the whole run)."""
        if self._fwd_in is None:
            self._fwd_in = (torch.empty_like(x), torch.empty_like(hidden))
        self._fwd_in[0].copy_(x)
        self._fwd_in[1].copy_(hidden)
        if self._fwd_graph is None:
            if self._fwd_warm < 2:
                self._fwd_warm += 1
                return self.model(*self._fwd_in)
            self._fwd_graph = torch.cuda.CUDAGraph()
            with gcapture(self._fwd_graph):
                self._fwd_out = self.model(*self._fwd_in)
        self._fwd_graph.replay()
        return self._fwd_out

    def compute_advantages(self):
        """:619-644 -> (adv_actor [T,N] un-normalised, returns [T,N])."""
        b, cfg = self.buffer, self.cfg
        return ops.gae_decoupled(b.rewards, b.values, b.dones, b.next_value, cfg.gamma, cfg.lam_actor, cfg.lam_critic)

    def update_model(self, advantages, returns):
        """:657-812.  Returns the metric means the reference prints."""
        cfg, b = self.cfg, self.buffer
        T, N, L, S, mb = b.T, b.N, int(cfg.seq_len), self.num_sequences, int(cfg.batch_size)
        states = b.states[:T].reshape(T * N, -1)
        hidden_flat = b.hidden_states.view(T * N, -1)
        act, lp, ent_old, val_old = b.actions.view(-1), b.log_probs.view(-1), b.old_entropies.view(-1), b.values.view(-1)
        adv, ret = advantages.reshape(-1).contiguous(), returns.reshape(-1).contiguous()
        steps = torch.arange(L, device=self.device, dtype=torch.int64)
        n_mb = S // mb
        metrics = torch.zeros(cfg.num_epochs * n_mb, 10, dtype=torch.float64, device=self.device)
        row = 0
        lcfg = (cfg.clip_eps_min, cfg.clip_eps_max, cfg.dual_clip, cfg.erc_beta_low, cfg.erc_beta_high, self.ent_coef)
        def minibatch(seq, metrics_row, rnd_out, bias=None):
            first = (seq // N) * (L * N) + seq % N                         # flat row of each window's first step
            rows = first.unsqueeze(1) + steps * N                          # [mb, L]
            idx = rows.reshape(-1).to(torch.int32)
            s_batch = states.index_select(0, rows.reshape(-1)).view(mb, L, -1)
            logits, values, _, predict, target = self.model(s_batch, hidden_flat.index_select(0, first))
            logits_flat, values_flat = logits.reshape(mb * L, -1), values.reshape(-1)
            mul = None
            if cfg.clip_cov_ratio > 0:                                     # :729-753 (off by default)
                li = idx.long()
                mul = cov_clip_mask(cfg, logits_flat.detach(), act.index_select(0, li), adv.index_select(0, li), self._perm_gen)
            dlogits, dvalues = ops.ppo_rnn_loss_fwd_bwd(logits_flat.detach(), values_flat.detach(), act, lp, ent_old,
                                                        val_old, adv, ret, lcfg, idx=idx, metrics_sum=metrics_row, corr_mul=mul)
            rnd_loss = (predict - target).pow(2).mean()                    # :775
            self._sink.arm()
            torch.autograd.backward([logits_flat, values_flat, rnd_loss], [dlogits, dvalues, None])
            self._sink.collect()
            if self.collective:
                gdist.all_reduce_sum(self.flat_grads)
            self.optimizer.step(grad_scale=1.0 / self.world_size, bias_dev=bias)
            rnd_out.copy_(rnd_loss.detach())

        # Same scheme as PPO-full: the minibatch body is captured once per update_model() call and replayed.
        graphed = (bool(getattr(cfg, "use_graphs", True)) and not self.collective and self.grad_norms is None
                   and cfg.num_epochs * n_mb > 2 and cfg.clip_cov_ratio <= 0)
        graph = None
        if graphed and self._g_seq is None:
            from .graphs import StepScalars
            self._scalars = StepScalars(self.device)
            self._g_bias, self._g_off = self._scalars.slot(16, torch.float32)
            self._g_seq = torch.empty(mb, dtype=torch.int64, device=self.device)
            self._g_row = torch.zeros(10, dtype=torch.float64, device=self.device)
            self._g_rnd = torch.zeros((), device=self.device)
        rnd_all = torch.zeros(cfg.num_epochs * n_mb, device=self.device)
        for _ in range(cfg.num_epochs):
            if self._parity_perms is not None:
                perm = next(self._parity_perms).to(self.device, torch.int64)
            else:
                perm = torch.randperm(S, device=self.device, generator=self._perm_gen)
            for start in range(0, S, mb):
                seq = perm[start:start + mb]
                if not graphed:
                    minibatch(seq, metrics[row], rnd_all[row])
                    if self.grad_norms is not None:
                        self.grad_norms.append(float(self.optimizer._sq.sqrt().item()))
                else:
                    self._g_seq.copy_(seq)
                    self._g_row.zero_()
                    self._scalars.set(self._g_off, self.optimizer.next_bias())
                    self._scalars.flush()
                    if self._g_warm < 2:
                        self._g_warm += 1
                        minibatch(self._g_seq, self._g_row, self._g_rnd, self._g_bias)
                    else:
                        if graph is None:
                            graph = torch.cuda.CUDAGraph()
                            with gcapture(graph):
                                minibatch(self._g_seq, self._g_row, self._g_rnd, self._g_bias)
                        graph.replay()
                    metrics[row].copy_(self._g_row)
                    rnd_all[row].copy_(self._g_rnd)
                row += 1
        del graph
        if cfg.anneal:                                                     # :794-800 (after the update)
            frac = 1 - self.step_count * self.world_size / cfg.max_train_steps
            self.lr = cfg.lr * frac
            for param_group in self.optimizer.param_groups:
                param_group["lr"] = self.lr
            self.ent_coef = cfg.entropy_coef * frac
        done = b.ep_returns[b.dones.bool()][-self.episode_rewards.maxlen:]
        for r in done.tolist():
            self.episode_rewards.append(r)
        m = metrics.cpu().numpy()
        self._last_metrics = m
        Bs = float(mb * L)
        cnt = np.where(m[:, 9] > 0, m[:, 9], 1.0)
        live = (m[:, 9] > 0).astype(np.float64)
        cov = (m[:, 8] - m[:, 6] * m[:, 7] / Bs) / Bs                      # covs.mean() per minibatch (:729-731)
        return {"policy_loss": float((m[:, 0] / cnt * live).mean()), "value_loss": float((m[:, 1] / cnt * live).mean()),
                "entropy": float((m[:, 2] / cnt * live).mean()), "clip_frac": float((m[:, 3] / cnt * live).mean()),
                "approx_kl": float((m[:, 4] / Bs).mean()), "erc_clip_frac": float((m[:, 5] / Bs).mean()),
                "cov": float(cov.mean()), "rnd_loss": float(rnd_all.mean().item())}

    def train(self):
        update_count = 0
        while self.step_count * self.world_size < self.cfg.max_train_steps:
            self.collect_experience()
            advantages, returns = self.compute_advantages()
            metrics = self.update_model(advantages, returns)
            update_count += 1
            if self.episode_rewards and self.rank == 0:
                print(f"Step: {self.step_count * self.world_size:,} | Updates: {update_count} | "
                      f"Avg Reward: {np.mean(self.episode_rewards):.1f} | KL: {metrics['approx_kl']:.4f}")
        self.env.close()

    def save_checkpoint(self, path):
        """Reference layout for recurrent agents (ppo_rnn_lunarlander.py:372-392 keys `net_state_dict`,
        `optimizer_state_dict`, `learn_step`) through the ModelLoader-style writer."""
        from .utils import checkpoint
        return checkpoint.save_agent(path, {"net": self.model}, {"optimizer": (self.model, self.optimizer)},
                                     learn_step=self.rollout_count, step_count=self.step_count,
                                     episode_rewards=list(self.episode_rewards), lr=self.lr, ent_coef=self.ent_coef)

    def load_checkpoint(self, path):
        from .utils import checkpoint
        rest = checkpoint.load_agent(path, {"net": self.model}, {"optimizer": (self.model, self.optimizer)})
        self.rollout_count = int(rest.get("learn_step", 0))
        self.step_count = int(rest.get("step_count", 0))
        self.lr, self.ent_coef = float(rest.get("lr", self.lr)), float(rest.get("ent_coef", self.ent_coef))
        self.episode_rewards.clear()
        self.episode_rewards.extend(rest.get("episode_rewards", []))
        gdist.broadcast(self.flat_params)
        return rest

    @torch.no_grad()
    def eval(self, num_episodes=10):
        """:820-846 as `num_episodes` parallel deterministic episodes, each with its own GRU state."""
        env = self._eval_env(num_episodes)
        obs = env.reset()
        nxt = torch.empty_like(obs)
        rew = torch.empty(num_episodes, device=self.device)
        done = torch.zeros(num_episodes, dtype=torch.uint8, device=self.device)
        ep_ret = torch.zeros(num_episodes, device=self.device)
        result = torch.full((num_episodes,), float("nan"), device=self.device)
        hidden = torch.zeros(num_episodes, self.hidden_size, device=self.device)
        for _ in range(env.max_steps + 1):
            act, _, _, hidden, _, _, _ = self.model.get_action(obs, hidden, deterministic=True)
            env.step(act, nxt, rew, done_out=done, ep_ret_out=ep_ret)
            result = torch.where(done.bool() & torch.isnan(result), ep_ret, result)
            obs, nxt = nxt, obs
            if not torch.isnan(result).any():
                break
        return result.tolist()

    def _eval_env(self, n):
        return VecEnv(self.cfg.env_name, n, device=self.device, seed=self.base_seed + 1_000_003, env_id0=1 << 40)

    def test(self):
        return self.eval(num_episodes=5)


if __name__ == "__main__":       # python -m gymrl_amd.ppo_lstm_lunarlander [--<Config attribute> <value> ...]  (ppo_lstm_lunarlander.py:886-903)
    from .utils.cli import run_script
    run_script(Config, PPOTrainer, interrupted="\nCtrl+C detected, stopping training and starting test...")
