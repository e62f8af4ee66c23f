"""PPO-full (mHC backbone, decoupled-lambda GAE, clip-higher, ratio-clamp dual clip, entropy-
ratio mask, LR + entropy annealing) — MI355X engine behind the reference's
algorithms/ppo_full_lunarlander.py surface: Config :19-52, ActorCritic :364-412,
RolloutBuffer :416-436, PPOTrainer :440-747 (collect_experience :462-505,
compute_advantages :507-535, update_model :537-679).

The mHC network (manifold hyper-connections with a 10-iteration Sinkhorn-Knopp on the
exp of a rate x rate mixing matrix) is dense PyTorch-ROCm autograd work — SURVEY.md section 8a
F1 keeps it off the hand-kernel list.  It is written here from the architecture's definition
with the reference's parameter names, so reference state_dicts load unchanged
(tests/test_trainers_gpu.py checks forward/backward equivalence on a golden).  Rollout,
categorical sampling (+ behaviour entropy), G3 GAE, the L3 loss forward/backward + metrics,
clip-norm + Adam and the gradient all-reduce are the HIP / RCCL path.
"""
import math
import os
import struct
from collections import deque

import numpy as np
import torch
import torch.nn as nn

from . import dist as gdist
from . import ops
from .graphs import capture as gcapture
from .envs import VecEnv
from .flat import FusedAdam, GradSink, flatten_module
from .nn import GradSink_direct, SmallLinear, skinny_matmul, smallk_linear, wide_linear_pair


class Config:
    def __init__(self):
        self.env_name = "LunarLander-v3"
        self.seed = None
        self.use_mhc = True
        self.mhc_dim = 128
        self.mhc_rate = 2
        self.mhc_layers = 2
        self.mhc_sk_it = 10
        self.persistent_rollout = True    # LunarLander + the default network shape: the rollout as one launch (gymrl_rollout_lunar_mhc)
        self.rollout_refill = True        # ... whose wave 1 prepares every env's next episode while wave 0 steps
        self.max_train_steps = 5e6
        self.update_freq = 4096            # steps PER ENV per rollout
        self.num_epochs = 4
        self.batch_size = 1024
        self.num_minibatches = None        # if set: minibatch = T*N / num_minibatches
        self.gamma = 0.995
        self.lam_actor = 0.95
        self.lam_critic = 0.95
        self.clip_eps_min = 0.2
        self.clip_eps_max = 0.28
        self.clip_cov_ratio = 0.0
        self.clip_cov_min = 1.0
        self.clip_cov_max = 5.0
        self.dual_clip = 3.0
        self.entropy_coef = 0.01
        self.erc_beta_low = 0.06
        self.erc_beta_high = 0.06
        self.lr = 3e-4
        self.max_grad_norm = 0.5
        self.anneal = True
        self.device = "cuda"
        self.num_envs = 1
        self.use_graphs = True             # replay the minibatch update as captured hipGraphs (equal minibatches)
        self.micro_batch = 0               # rows per forward/backward pass (0: the whole minibatch); a minibatch larger than
                                           # this is accumulated over equal micro-batches before ONE optimiser step — the
                                           # same gradient, bounded activation memory (4096 envs x F0's ratio = 4.2 M rows)
        self.gae_variant = 1               # 1 = time-blocked G3 with the chunk maps composed during the rollout, 0 = sequential


FUSED_GATES = True          # training pass: the gates as one forward + one backward launch (False: torch ops + gymrl_sinkhorn)
FUSED_MIXING = True         # training pass: read / combine products as fused forward + backward launches (False: broadcast multiplies)
FUSED_INFERENCE = True      # rollout forward on the inference kernels (tools A/B; False: the torch modules)
FUSED_POLICY = True         # rollout forward as ONE launch (gymrl_mhc_policy_forward) when the network has the default shape
POLICY_IMAGE = True         # ... the persistent rollout reading its wide weights from a packed image (gymrl_mhc_policy_pack, once per rollout)
FUSED_SUB_FORWARD = True    # ... and its forward as ONE launch when D = 128 (False: gates + Linear + combine launches)
FUSED_HEAD_TAIL = True      # training pass: a head's SiLU -> RMSNorm -> output Linear as one launch each way
KEEP_UPDATE_GRAPHS = os.environ.get("GYMRL_KEEP_UPDATE_GRAPHS", "1") != "0"   # update_model(): captured graphs outlive the call
                            # (and with them their private memory pool — the activations of one micro-batch, several GB at 524 288
                            # rows — stays allocated through the rollout phase; "0" re-captures per call and frees it: +19 ms per update)
FUSED_HEAD_PAIR = True      # training pass: actor.mlp.0 and critic.mlp.0 as one autograd node (their input gradients added in the GEMM)
FUSED_SUB_BACKWARD = True   # ... and its backward as ONE launch + the Linear's weight gradient (False: the five backward launches)
FUSED_SUB = True            # training pass: a whole hyper-connection sub-block as one autograd node (3 launches forward, 7 backward)
FUSED_NORM = True           # training pass: RMSNorm (+ the SiLU before it) as one launch each way


def _ortho(layer, std):
    nn.init.orthogonal_(layer.weight, gain=std)
    if layer.bias is not None:
        nn.init.constant_(layer.bias, 0)
    return layer


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x, silu=False):
        """silu=True: the norm of SiLU(x) (the MLPs' Linear -> SiLU -> RMSNorm)."""
        if (FUSED_NORM and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] <= 512 and x.shape[0] > 0):
            return _RmsNorm.apply(x, self.weight, self.eps, silu)          # one launch each way (~20 through torch)
        if silu:
            x = torch.nn.functional.silu(x)
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight


class _RmsNorm(torch.autograd.Function):
    """RMSNorm (:96-104), optionally of SiLU(x), as gymrl_rmsnorm / gymrl_rmsnorm_bwd."""

    @staticmethod
    def forward(ctx, x, w, eps, silu):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.eps, ctx.act = eps, ops.LIN_ACT["silu"] if silu else 0
        return ops.rmsnorm(x, w, eps, act=ctx.act)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        d_x, d_w = ops.rmsnorm_bwd(g.contiguous(), x, w, ctx.eps, ctx.act)
        return d_x, d_w, None, None


class _RmsNormSum(torch.autograd.Function):
    """final_norm(h.sum(dim=1)) (:243) as one launch each way: the branch sum is taken on load, and the backward hands the branches
    ONE [B, D] gradient as a stride-0 view — _MhcSub.backward reads it as such, nothing [B, n, D] is written or re-read."""

    @staticmethod
    def forward(ctx, h, w, eps):
        h = h.contiguous()
        ctx.save_for_backward(h, w)
        ctx.eps = eps
        return ops.rmsnorm(h, w, eps, n_sum=h.shape[1])

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        d_x, d_w = ops.rmsnorm_bwd(g.contiguous(), h, w, ctx.eps, 0, n_sum=h.shape[1])
        return d_x.unsqueeze(1).expand(-1, h.shape[1], -1), d_w, None


class _NormProj(torch.autograd.Function):
    """A head's tail SiLU -> RMSNorm -> Linear(D -> n_out) (:371-402) as gymrl_norm_proj_fwd / _bwd: the normalised activations
    and their gradient stay in registers (as three nodes they are 2.2 KB per row of HBM traffic for 16 bytes of output)."""

    @staticmethod
    def forward(ctx, x, norm_w, eps, W2, b2):
        x = x.contiguous()
        ctx.save_for_backward(x, norm_w, W2)
        ctx.eps, ctx.has_bias = eps, b2 is not None
        return ops.norm_proj_fwd(x, norm_w, eps, W2, b2)

    @staticmethod
    def backward(ctx, g):
        x, norm_w, W2 = ctx.saved_tensors
        d_x, d_nw, d_W2, d_b2 = ops.norm_proj_bwd(g.contiguous(), x, norm_w, ctx.eps, W2)
        return d_x, d_nw, None, d_W2, d_b2 if ctx.has_bias else None


class ManifoldHyperConnectionFuse(nn.Module):
    """One hyper-connection: per-sample gates (H_pre, H_post) and a doubly-stochastic branch
    mixing matrix H_res from an RMS-fused linear read-out of the branch stack (:106-194)."""

    def __init__(self, dim, rate, max_sk_it):
        super().__init__()
        self.n, self.dim, self.nc, self.max_sk_it = rate, dim, rate * dim, max_sk_it
        n = rate
        self.norm = RMSNorm(dim * rate)
        self.w = nn.Parameter(torch.zeros(self.nc, n * n + 2 * n))
        self.alpha = nn.Parameter(torch.ones(3) * 0.01)
        beta = torch.zeros(n * n + 2 * n)
        beta[:2 * n] = 0.01
        mix = torch.full((n, n), -2.0)
        mix.fill_diagonal_(2.0)                      # start close to the identity: branches stay independent
        beta[2 * n:] = mix.flatten()
        self.beta = nn.Parameter(beta)

    def gates(self, h):
        """h [B, n, D] -> (pre [B, n], post [B, n], mix [B, n, n])."""
        B, n = h.shape[0], self.n
        if h.is_cuda and FUSED_GATES and n == 2 and self.nc in (256, 512) and h.dtype == torch.float32:
            return _MhcGates.apply(h, self.norm.weight, self.w, self.alpha, self.beta, self.max_sk_it)
        flat = h.reshape(B, self.nc)
        H = skinny_matmul(self.norm.weight * flat, self.w)       # [B, nc] x [nc, n*n + 2n]: HBM-bound kernels (gymrl_amd/nn.py)
        r_inv = 1.0 / (flat.norm(dim=-1, keepdim=True) / math.sqrt(self.nc) + 1e-6)
        pre = torch.sigmoid(r_inv * H[:, :n] * self.alpha[0] + self.beta[:n])
        post = 2 * torch.sigmoid(r_inv * H[:, n:2 * n] * self.alpha[1] + self.beta[n:2 * n])
        A = (r_inv * H[:, 2 * n:] * self.alpha[2] + self.beta[2 * n:]).reshape(B, n, n).exp()
        if A.is_cuda and n in (2, 4):                # the sweeps in one launch (~6 launches per sweep through torch)
            u, v = ops.sinkhorn(A.detach().contiguous(), self.max_sk_it)
            return pre, post, u.unsqueeze(2) * A * v.unsqueeze(1)
        with torch.no_grad():                        # Sinkhorn-Knopp scalings, treated as constants
            u = torch.ones(B, n, device=h.device)
            v = torch.ones(B, n, device=h.device)
            for _ in range(self.max_sk_it):
                u = 1.0 / ((A * v.unsqueeze(1)).sum(-1) + 1e-8)
                v = 1.0 / ((A * u.unsqueeze(2)).sum(1) + 1e-8)
        return pre, post, u.unsqueeze(2) * A * v.unsqueeze(1)


class _MhcGates(torch.autograd.Function):
    """ManifoldHyperConnectionFuse.gates (:125-147) as one launch forward (gymrl_mhc_gates: read-out, sigmoids, exp, Sinkhorn
    sweeps) and one backward (gymrl_mhc_gates_bwd: the Sinkhorn scalings are constants, as under the reference's no_grad)."""

    @staticmethod
    def forward(ctx, h, norm_w, w, alpha, beta, sk_it):
        h = h.contiguous()
        pre, post, mix, _, stats = ops.mhc_gates(h, norm_w, w, alpha, beta, sk_it, stats=True)
        ctx.save_for_backward(h, norm_w, w, alpha, pre, post, mix, stats)
        return pre, post, mix

    @staticmethod
    def backward(ctx, d_pre, d_post, d_mix):
        h, norm_w, w, alpha, pre, post, mix, stats = ctx.saved_tensors
        z = lambda g, ref: torch.zeros_like(ref) if g is None else g.contiguous()   # noqa: E731
        d_h, d_nw, d_w, d_alpha, d_beta = ops.mhc_gates_bwd(h, norm_w, w, alpha, pre, post, mix, stats, z(d_pre, pre),
                                                            z(d_post, post), z(d_mix, mix))
        return d_h, d_nw, d_w, d_alpha, d_beta, None


_LIBRARY_ROWS = 16384       # rows from which a 128-wide product leaves the one-wave-per-tile layer kernels for the weight-stationary
                            # GEMM kernels of csrc/gemm.hip (forward / input gradient; other widths: the library)


class _MhcSub(torch.autograd.Function):
    """One hyper-connection sub-block h -> post (x) SiLU(Linear(sum_i pre_i h_i)) + mix h (MHCBlock._sub :160-165 with the gates
    :125-147) as ONE autograd node.  Forward: gymrl_mhc_gates (gates + read-out sums + branch sum), gymrl_lin_fwd, gymrl_mhc_combine
    (SiLU applied on load).  Backward: combine -> (d post, d mix, d z), the Linear's two gradient launches, the read's d pre, and
    gymrl_mhc_gates_bwd, which also folds the read's and the combine's paths into d h: the three consumers of h hand autograd one
    gradient tensor (as separate nodes they cost two [B, n, D] additions, the SiLU's two passes and a second read of h)."""

    @staticmethod
    def forward(ctx, h, norm_w, w, alpha, beta, W, b, sk_it):
        h = h.contiguous()
        ctx.sinks = (getattr(W, "_gymrl_sink", None), getattr(b, "_gymrl_sink", None))
        ctx.repeated = h.dim() == 2                          # [B, D]: the input projection, the same row for both branches
        if ctx.repeated and not (FUSED_SUB_FORWARD and FUSED_SUB_BACKWARD and h.shape[1] == 128):
            raise ValueError("_MhcSub takes a [B, D] input only on the one-launch kernels (D = 128)")
        if FUSED_SUB_FORWARD and h.shape[-1] == 128:         # gates + Linear + combine in one launch (csrc/mhc.hip mhc_sub_fwd_kernel)
            pre, post, mix, stats, read, z, h_out = ops.mhc_sub_forward(h, norm_w, w, alpha, beta, W, b, sk_it)
            ctx.save_for_backward(h, norm_w, w, alpha, pre, post, mix, stats, read, z, W)
            return h_out
        pre, post, mix, read, stats = ops.mhc_gates(h, norm_w, w, alpha, beta, sk_it, stats=True)
        # the Linear: the layer kernels below _LIBRARY_ROWS rows (one launch, bias inside), the library GEMM above (97 us at
        # 262144 x 128 x 128 against 256: profiles/r02_micro_lin_large.json)
        if h.shape[0] >= _LIBRARY_ROWS and ops.linear_shape_ok(W.shape[1], W.shape[0]):
            z = ops.linear_fwd(read, W, b, torch.empty(read.shape[0], W.shape[0], device=read.device), act=False)
        elif h.shape[0] >= _LIBRARY_ROWS:
            z = torch.addmm(b, read, W.t())
        else:
            z = ops.lin_fwd(read, W, b)
        ctx.save_for_backward(h, norm_w, w, alpha, pre, post, mix, stats, read, z, W)
        return ops.mhc_combine(post, mix, z, h, act=ops.LIN_ACT["silu"])

    @staticmethod
    def backward(ctx, g):
        h, norm_w, w, alpha, pre, post, mix, stats, read, z, W = ctx.saved_tensors
        if FUSED_SUB_BACKWARD and h.shape[-1] == 128:        # csrc/mhc.hip mhc_sub_bwd_kernel: g and h cross HBM once
            # a stride-0 branch dimension is _RmsNormSum's gradient: one [B, D] row for both branches, read as such
            g = g[:, 0].contiguous() if g.stride(1) == 0 else g.contiguous()
            d_z, d_h, d_nw, d_w, d_alpha, d_beta = ops.mhc_sub_backward(g, h, z, pre, post, mix, stats, norm_w, w, alpha, W,
                                                                        sum_branches=ctx.repeated)
            slot = GradSink_direct(ctx.sinks[0], ctx.sinks[1], True)
            if slot is not None:
                ops.lin_bwd_weight(d_z, z, read, slot[0], slot[1], accumulate=slot[2])
                d_W = d_b = None
            else:
                d_W, d_b = torch.empty_like(W), torch.empty(W.shape[0], dtype=W.dtype, device=W.device)
                ops.lin_bwd_weight(d_z, z, read, d_W, d_b)
            return d_h, d_nw, d_w, d_alpha, d_beta, d_W, d_b, None
        g = g.contiguous()
        d_post, d_mix, d_z, _ = ops.mhc_combine_bwd(g, post, mix, z, h, act=ops.LIN_ACT["silu"], want_dh=False)
        if h.shape[0] >= _LIBRARY_ROWS and ops.linear_shape_ok(W.shape[1], W.shape[0]):
            d_read = ops.linear_bwd_input(d_z, W, None, torch.empty_like(read))      # csrc/gemm.hip, exact f32 MFMA
        elif h.shape[0] >= _LIBRARY_ROWS:
            d_read = torch.mm(d_z, W)
        else:
            d_read, _ = ops.lin_bwd_input(d_z, z, W)
        slot = GradSink_direct(ctx.sinks[0], ctx.sinks[1], True)
        if slot is not None:                         # straight into the flat gradient buffer (gymrl_amd/flat.py GradSink)
            ops.lin_bwd_weight(d_z, z, read, slot[0], slot[1], accumulate=slot[2])
            d_W = d_b = None
        else:
            d_W, d_b = torch.empty_like(W), torch.empty(W.shape[0], dtype=W.dtype, device=W.device)
            ops.lin_bwd_weight(d_z, z, read, d_W, d_b)
        d_pre, _ = ops.mhc_read_bwd(d_read, pre, h, want_dh=False)
        d_h, d_nw, d_w, d_alpha, d_beta = ops.mhc_gates_bwd(h, norm_w, w, alpha, pre, post, mix, stats, d_pre, d_post, d_mix,
                                                            d_read=d_read, g_out=g)
        return d_h, d_nw, d_w, d_alpha, d_beta, d_W, d_b, None


class _MhcRead(torch.autograd.Function):
    """read = sum_i pre_i h_i (:161) with its backward, one launch each way (csrc/mhc.hip)."""

    @staticmethod
    def forward(ctx, pre, h):
        pre, h = pre.contiguous(), h.contiguous()
        ctx.save_for_backward(pre, h)
        return ops.mhc_read_fwd(pre, h)

    @staticmethod
    def backward(ctx, g):
        pre, h = ctx.saved_tensors
        return ops.mhc_read_bwd(g.contiguous(), pre, h)


class _MhcCombine(torch.autograd.Function):
    """h' = post (x) out + mix h (:165) with its backward, one launch each way."""

    @staticmethod
    def forward(ctx, post, mix, out, h):
        post, mix, out, h = post.contiguous(), mix.contiguous(), out.contiguous(), h.contiguous()
        ctx.save_for_backward(post, mix, out, h)
        return ops.mhc_combine(post, mix, out, h)

    @staticmethod
    def backward(ctx, g):
        post, mix, out, h = ctx.saved_tensors
        return ops.mhc_combine_bwd(g.contiguous(), post, mix, out, h)


class MHCBlock(nn.Module):
    def __init__(self, dim, rate, max_sk_it):
        super().__init__()
        self.linear1 = SmallLinear(dim, dim)
        self.mhc1 = ManifoldHyperConnectionFuse(dim, rate, max_sk_it)
        self.linear2 = SmallLinear(dim, dim)
        self.mhc2 = ManifoldHyperConnectionFuse(dim, rate, max_sk_it)
        self.act = nn.SiLU()

    @staticmethod
    def _sub(h, fuse, linear, act):
        if h.dim() == 2 and not (FUSED_SUB and FUSED_GATES and FUSED_MIXING and FUSED_SUB_FORWARD and FUSED_SUB_BACKWARD
                                 and h.shape[1] == 128 and fuse.n == 2):
            h = h.unsqueeze(1).repeat(1, fuse.n, 1)          # (only the one-launch node takes the un-repeated input projection)
        if (FUSED_SUB and FUSED_GATES and FUSED_MIXING and h.is_cuda and h.dtype == torch.float32 and fuse.n == 2
                and fuse.nc in (256, 512) and isinstance(act, nn.SiLU) and isinstance(linear, SmallLinear)
                and linear.bias is not None and getattr(linear, "act", None) in (None, "none")):
            return _MhcSub.apply(h, fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta, linear.weight, linear.bias, fuse.max_sk_it)
        pre, post, mix = fuse.gates(h)
        if h.is_cuda and h.shape[1] in (2, 4) and h.shape[2] % 4 == 0 and FUSED_MIXING:
            # the three products and their backward as four launches instead of ~25 elementwise passes over [B, n, D]
            out = act(linear(_MhcRead.apply(pre, h)))
            return _MhcCombine.apply(post, mix, out, h)
        if h.is_cuda:
            # the same three products as broadcast multiplies: the library answers 262144 batched 1 x n x D GEMMs (one per
            # row of a micro-batch) in 2.5-3.6 ms each way — 78 % of a PPO-full update (`profiles/r02_ppo_full_kernel_stats.csv`)
            # — where the operands are ~0.1 ms of HBM time
            read = (pre.unsqueeze(2) * h).sum(1)                               # weighted sum of branches  [B, D]
            out = act(linear(read)).unsqueeze(1)                               # (2-D: the layer kernels take it)
            mixed = mix[:, :, 0:1] * h[:, 0:1, :]
            for j in range(1, h.shape[1]):
                mixed = mixed + mix[:, :, j:j + 1] * h[:, j:j + 1, :]
            return post.unsqueeze(2) * out + mixed                             # broadcast back + inter-branch mixing
        read = torch.bmm(pre.unsqueeze(1), h)                    # weighted sum of branches  [B, 1, D]
        out = act(linear(read))
        return torch.bmm(post.unsqueeze(2), out) + torch.bmm(mix, h)   # broadcast back + inter-branch mixing

    def forward(self, h):
        h = self._sub(h, self.mhc1, self.linear1, self.act)
        return self._sub(h, self.mhc2, self.linear2, self.act)


class MHCBackbone(nn.Module):
    def __init__(self, input_dim, output_dim, rate, num_layers, max_sk_it):
        super().__init__()
        self.rate = rate
        self.input_proj = SmallLinear(input_dim, output_dim)
        self.layers = nn.ModuleList([MHCBlock(output_dim, rate, max_sk_it) for _ in range(num_layers)])
        self.final_norm = RMSNorm(output_dim)

    def forward(self, x):
        z0 = smallk_linear(x, self.input_proj) if FUSED_SUB and x.is_cuda and x.dim() == 2 else None
        if z0 is None:
            z0 = self.input_proj(x)
        # training pass on the one-launch sub-block kernels: the first sub-block reads the projection as the repeated row it is
        # and returns the branches' summed gradient, the final norm sums the branches on load and hands both ONE gradient row —
        # the reference's repeat / sum (:239, :243) and their backward cost four [B, n, D] passes through torch (0.38 ms per
        # 262144-row micro-batch, `profiles/r03_ppo_full_kernel_stats.csv`)
        direct = (FUSED_SUB and FUSED_GATES and FUSED_MIXING and FUSED_SUB_FORWARD and FUSED_SUB_BACKWARD and FUSED_NORM
                  and z0.is_cuda and z0.dtype == torch.float32 and self.rate == 2 and z0.shape[1] == 128 and z0.shape[0] > 0
                  and len(self.layers) > 0)
        h = z0 if direct else z0.unsqueeze(1).repeat(1, self.rate, 1)     # [B, n, D]
        for layer in self.layers:
            h = layer(h)
        if direct:
            return _RmsNormSum.apply(h, self.final_norm.weight, self.final_norm.eps)
        return self.final_norm(h.sum(dim=1))


@torch.no_grad()
def cov_clip_mask(cfg, logits, act_b, adv_b, generator=None, perm=None):
    """Covariance clip of ppo_full_lunarlander.py:594-616 (ppo_lstm :729-753): among the minibatch rows whose
    (log-prob - mean) * (advantage - mean) lies in (clip_cov_min, clip_cov_max), a random clip_cov_ratio of them
    (at least one) is taken out of every masked mean.  Returns the f32[B] multiplier the loss kernels take as
    `corr_mul`, or None when the branch is off.  `perm` replays an explicit torch.randperm (parity tests)."""
    if cfg.clip_cov_ratio <= 0:
        return None
    lp = torch.log_softmax(logits, dim=-1).gather(1, act_b.long().unsqueeze(1)).squeeze(1)
    covs = (lp - lp.mean()) * (adv_b - adv_b.mean())
    cand = torch.where((covs > cfg.clip_cov_min) & (covs < cfg.clip_cov_max))[0]     # one host read: the branch is optional
    mul = torch.ones_like(adv_b)
    n = int(cand.numel())
    if n > 0:
        k = min(max(int(n * cfg.clip_cov_ratio), 1), n)
        if perm is None:
            perm = torch.randperm(n, device=logits.device, generator=generator)
        mul[cand[torch.as_tensor(perm, device=logits.device).long()[:k]]] = 0.0
    return mul


class MLP(nn.Module):
    """Linear -> SiLU -> RMSNorm -> ... -> Linear [-> SiLU -> RMSNorm when last_act]; keys `mlp.<i>` (:371-402)."""

    def __init__(self, dims, last_act=False, last_std=None):
        super().__init__()
        layers = []
        for i in range(len(dims) - 1):
            last = i == len(dims) - 2
            layers.append(_ortho(SmallLinear(dims[i], dims[i + 1]), last_std if (last and last_std) else np.sqrt(2)))
            if not last or last_act:
                layers += [nn.SiLU(), RMSNorm(dims[i + 1])]
        self.mlp = nn.Sequential(*layers)

    def forward(self, x, first=None):
        """first: the output of mlp.0 computed by the caller (ActorCritic: both heads' first layers as one autograd node)."""
        mods = list(self.mlp)
        i = 0
        if first is not None:
            x, i = first, 1
        while i < len(mods):
            if (FUSED_HEAD_TAIL and FUSED_NORM and i + 3 == len(mods) and isinstance(mods[i], nn.SiLU) and isinstance(mods[i + 1], RMSNorm)
                    and isinstance(mods[i + 2], SmallLinear) and getattr(mods[i + 2], "act", None) in (None, "none")
                    and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] > 0
                    and ops.norm_proj_ok(x.shape[1], mods[i + 2].out_features)):
                # SiLU -> RMSNorm -> the output Linear in one launch each way (csrc/mhc.hip norm_proj_*_kernel)
                return _NormProj.apply(x, mods[i + 1].weight, mods[i + 1].eps, mods[i + 2].weight, mods[i + 2].bias)
            if i + 1 < len(mods) and isinstance(mods[i], nn.SiLU) and isinstance(mods[i + 1], RMSNorm):
                x = mods[i + 1](x, silu=True)        # SiLU rides in the norm's launches on the GPU
                i += 2
            else:
                x = mods[i](x)
                i += 1
        return x


class PSCN(nn.Module):
    """Parallel split-and-concatenate tower: layer i maps to width/2^i, half of its output is emitted and the
    other half feeds layer i+1 (:405-446)."""

    def __init__(self, input_dim, output_dim, depth):
        super().__init__()
        min_dim = 2 ** (depth - 1)
        if depth < 1 or output_dim < min_dim or output_dim % min_dim:
            raise ValueError("PSCN: output_dim must be a multiple of 2^(depth-1)")
        self.output_dim = output_dim
        self.layers = nn.ModuleList()
        in_dim, out_dim = input_dim, output_dim
        for _ in range(depth):
            self.layers.append(MLP([in_dim, out_dim], last_act=True))
            in_dim = out_dim // 2
            out_dim //= 2

    def forward(self, x):
        parts = []
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < len(self.layers) - 1:
                half = self.output_dim // (2 ** (i + 1))
                parts.append(x[..., :half])
                x = x[..., half:]
            else:
                parts.append(x)
        return torch.cat(parts, dim=-1)


class ActorCritic(nn.Module):
    def __init__(self, state_dim, action_dim, config=None):
        super().__init__()
        cfg = config or Config()
        if getattr(cfg, "use_mhc", True):
            self.shared = MHCBackbone(state_dim, cfg.mhc_dim, cfg.mhc_rate, cfg.mhc_layers, cfg.mhc_sk_it)
            shared_out = cfg.mhc_dim
        else:                                                     # :377-384
            self.shared = PSCN(state_dim, 256, 4)
            shared_out = 256
        self.actor = MLP([shared_out, 256, action_dim], last_std=0.001)
        self.critic = MLP([shared_out, 256, 1], last_std=1.0)

    def forward(self, x):
        x = self.shared(x)
        pair = wide_linear_pair(x, self.actor.mlp[0], self.critic.mlp[0]) if FUSED_HEAD_PAIR and x.is_cuda and x.dim() == 2 else None
        if pair is not None:        # the two heads' first layers share their input: ONE autograd node, no gradient add pass
            return self.actor(x, first=pair[0]), self.critic(x, first=pair[1])
        return self.actor(x), self.critic(x)

    @torch.no_grad()
    def forward_fused(self, x):
        """forward(x) without gradients on the inference kernels (csrc/mhc.hip, csrc/lin.hip): ~20 launches instead of
        ~400 — one per Linear (+ SiLU), three per hyper-connection (gates + read, Linear + SiLU, combine), the RMSNorms.
        Same values as the modules to 1e-5 (tests/test_mhc_fused_gpu.py).  None when the network is not the mHC one."""
        bb = self.shared
        if not isinstance(bb, MHCBackbone) or not x.is_cuda or bb.rate not in (2, 4):
            return None
        silu = ops.LIN_ACT["silu"]
        z = ops.lin_fwd(x.contiguous(), bb.input_proj.weight, bb.input_proj.bias)
        h = z.unsqueeze(1).repeat(1, bb.rate, 1)
        for layer in bb.layers:
            for fuse, linear in ((layer.mhc1, layer.linear1), (layer.mhc2, layer.linear2)):
                _, post, mix, read = ops.mhc_gates(h, fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta, fuse.max_sk_it)
                h = ops.mhc_combine(post, mix, ops.lin_fwd(read, linear.weight, linear.bias, silu), h)
        feat = ops.rmsnorm(h, bb.final_norm.weight, bb.final_norm.eps, n_sum=bb.rate)
        a, c = self.actor.mlp, self.critic.mlp
        if len(a) != 4 or len(c) != 4:
            return None
        ha, hc = ops.lin_fwd([feat, feat], [a[0].weight, c[0].weight], [a[0].bias, c[0].bias], silu)
        ha, hc = ops.rmsnorm(ha, a[2].weight, a[2].eps), ops.rmsnorm(hc, c[2].weight, c[2].eps)
        return ops.lin_fwd(ha, a[3].weight, a[3].bias), ops.lin_fwd(hc, c[3].weight, c[3].bias)

    def _policy_desc(self):
        """The gymrl_mhc_policy descriptor of this network (None when its shape is not the one-launch kernel's: n = 2 branches
        of 128, 256-wide heads, <= 16 observations, <= 8 actions, <= 8 sub-blocks), rebuilt when a parameter moved."""
        bb = self.shared
        a, c = self.actor.mlp, self.critic.mlp
        if (not isinstance(bb, MHCBackbone) or bb.rate != 2 or len(a) != 4 or len(c) != 4 or len(bb.layers) > 4
                or tuple(bb.input_proj.weight.shape) != (128, bb.input_proj.in_features) or bb.input_proj.in_features > 16
                or tuple(a[0].weight.shape) != (256, 128) or tuple(c[0].weight.shape) != (256, 128) or a[3].out_features > 8
                or c[3].out_features != 1 or bb.input_proj.bias is None):
            return None
        params = [bb.input_proj.weight, bb.input_proj.bias, bb.final_norm.weight]
        for layer in bb.layers:
            for fuse, linear in ((layer.mhc1, layer.linear1), (layer.mhc2, layer.linear2)):
                if linear.bias is None or fuse.nc != 256 or fuse.max_sk_it != bb.layers[0].mhc1.max_sk_it:
                    return None
                params += [fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta, linear.weight, linear.bias]
        for m in (a, c):
            if m[0].bias is None or m[3].bias is None:
                return None
            params += [m[0].weight, m[0].bias, m[2].weight, m[3].weight, m[3].bias]
        if any(p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() for p in params):
            return None
        key = tuple(p.data_ptr() for p in params)
        if getattr(self, "_desc_key", None) != key:
            from ._lib import MhcPolicy
            d = MhcPolicy()
            d.obs_dim, d.n_sub, d.n_act = bb.input_proj.in_features, 2 * len(bb.layers), a[3].out_features
            d.sk_it = bb.layers[0].mhc1.max_sk_it if len(bb.layers) else 0
            it = iter(key)
            d.in_w, d.in_b, d.final_norm_w, d.final_norm_eps = next(it), next(it), next(it), bb.final_norm.eps
            for s in range(d.n_sub):
                for f in ("norm_w", "w", "alpha", "beta", "lin_w", "lin_b"):
                    setattr(d.sub[s], f, next(it))
            for h, m in enumerate((a, c)):
                for f in ("w1", "b1", "norm_w", "w2", "b2"):
                    setattr(d.head[h], f, next(it))
                d.head[h].norm_eps = m[2].eps
            self._desc, self._desc_key = d, key
        return self._desc

    @torch.no_grad()
    def forward_policy(self, x, logits_out=None, value_out=None):
        """forward(x) without gradients as ONE launch (gymrl_mhc_policy_forward: 16 rows per workgroup through every layer);
        None when the network is not the shape that kernel is written for (then: forward_fused)."""
        if not (FUSED_POLICY and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()):
            return None
        d = self._policy_desc()
        if d is None:
            return None
        d.image = None                       # (an image is only as fresh as its last pack: this call reads the parameters in place)
        logits, value = ops.mhc_policy(d, x, logits_out, value_out)
        return logits, value.view(-1, 1)

    def forward_inference(self, x):
        """The rollout forward on the hand-written kernels: one launch when the shape allows, else one per layer; None: neither."""
        out = self.forward_policy(x)
        return self.forward_fused(x) if out is None else out

    @torch.no_grad()
    def get_action(self, x, deterministic=False, seed=0, counter=0, env_id0=0):
        """:395-407 batched -> (action i32[N], logp[N], value[N], entropy[N])."""
        out = self.forward_inference(x) if FUSED_INFERENCE else None
        logits, value = self.forward(x) if out is None else out
        act, logp, ent, val = ops.categorical_sample(logits, value=value.view(-1), seed=seed, counter=counter,
                                                     env_id0=env_id0, deterministic=deterministic)
        return act, logp, val, ent

    @torch.no_grad()
    def get_value(self, x):
        out = self.forward_inference(x) if FUSED_INFERENCE else None
        return (self.forward(x) if out is None else out)[1].view(-1)


class RolloutBuffer:
    """[T][N] slabs incl. the behaviour-policy entropies (:416-436)."""

    def __init__(self, T, N, obs_dim, device):
        z = lambda *s, **k: torch.zeros(*s, device=device, **k)   # noqa: E731
        self.T, self.N = T, N
        self.states = z(T + 1, N, obs_dim)
        self.actions = z(T, N, dtype=torch.int32)
        self.log_probs, self.values, self.rewards, self.old_entropies = z(T, N), z(T, N), z(T, N), z(T, N)
        self.dones = z(T, N, dtype=torch.uint8)
        self.ep_returns = z(T, N)
        self.next_value = z(N)

    def clear(self):
        pass


class PPOTrainer:
    def __init__(self, config):
        self.cfg = config
        if not torch.cuda.is_available() or not ops.device_ok():
            raise RuntimeError("gymrl_amd PPO-full needs an MI355X and libgymrl_hip.so; no CPU fallback")
        self.rank, self.world_size = gdist.rank(), gdist.world_size()
        self.collective = gdist.collectives_active()          # world_size > 1 (or one rank under GYMRL_FORCE_COLLECTIVES)
        self.device = torch.device(config.device if ":" in str(config.device) else f"cuda:{torch.cuda.current_device()}")
        self.base_seed = 0 if config.seed is None else int(config.seed)
        N = int(config.num_envs)
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
        self.step_count = 0
        self.rollout_count = 0
        self.episode_rewards = deque(maxlen=10)
        self.lr, self.ent_coef = config.lr, config.entropy_coef
        self.buffer = RolloutBuffer(int(config.update_freq), N, state_dim, self.device)
        self._perm_gen = torch.Generator(device=self.device)            # covariance-clip row draws (:607)
        self._perm_gen.manual_seed(self.base_seed * 7919 + 17 + self.rank)
        self._perm_seed, self._perm_draws, self._perm = self.base_seed * 7919 + 17 + self.rank, 0, None   # epoch shuffles
        self._parity_noise = None      # tests: list of f32[T, N, A] Exp(1) draws, one per rollout (popped)
        self._parity_perms = None      # tests: iterator of i32[T*N] shuffle orders, one per epoch (DataLoader's RandomSampler)
        self.grad_norms = None         # tests: set to [] to record the pre-clip gradient norm of every minibatch
        self._sink = GradSink(self.model)
        T = int(config.update_freq)
        self._gae_ws = ops.gae_decoupled_workspace(T, N, self.device)
        self._gae_run = torch.zeros(2, 2, N, dtype=torch.float64, device=self.device)   # running maps: actor, critic
        self._adv = torch.empty(T, N, device=self.device)
        self._ret = torch.empty(T, N, device=self.device)
        self._agg_ready = False
        self._g_idx, self._g_warm = None, 0          # hipGraph replay of the minibatch body (update_model)
        self._fwd_graph, self._fwd_in, self._fwd_out, self._fwd_warm = None, None, None, 0
        self._pol_out = None
        self._reducer, self._time_collectives = None, False     # gdist.GradReducer (world_size > 1); bench.py times it

    @torch.no_grad()
    def collect_experience(self):
        """:462-505 for N envs."""
        b, env, cfg = self.buffer, self.env, self.cfg
        seed = (self.base_seed if cfg.seed is not None else self.base_seed + 0x9E3779B1 * (self.rollout_count + 1))
        env.reset(b.states[0], seed=seed & 0x7FFFFFFFFFFFFFFF)
        c0 = self.rollout_count * b.T
        graphed = bool(getattr(cfg, "use_graphs", True))
        noise = self._parity_noise.pop(0) if self._parity_noise else None
        fuse_gae = getattr(cfg, "gae_variant", 1) == 1 and b.N % 4 == 0

        def online(t_prev):      # fold step t_prev (whose delta needs V_{t_prev + 1}) into its chunk's two affine maps
            return ops.gae_online(b.rewards[t_prev], b.dones[t_prev], b.values[t_prev], self._gae_run[0], self._gae_ws,
                                  t_prev, b.T, cfg.gamma, cfg.lam_actor, cfg.lam_critic, self._gae_run[1])
        # the whole forward as one launch when the network has the kernel's shape (the parameters do not move during a rollout:
        # one descriptor, fixed output buffers — nothing for a graph to save)
        desc = self.model._policy_desc() if (FUSED_INFERENCE and FUSED_POLICY and b.states.is_cuda) else None
        if desc is not None:
            desc.image = None                # (only the persistent launch below packs one, for its own T steps)
        if desc is not None and self._pol_out is None:
            self._pol_out = (torch.empty(b.N, desc.n_act, device=b.states.device), torch.empty(b.N, device=b.states.device))
        if (desc is not None and getattr(cfg, "persistent_rollout", True) and isinstance(env, VecEnv)
                and env.kind == ops.LUNARLANDER and desc.obs_dim == 8 and desc.n_act == 4):
            # the whole rollout as ONE launch: a workgroup owns 16 envs for all T steps (policy tile, draw, both GAE chunk maps,
            # Box2D step, slab writes) and never waits for another one — a step costs the mean wave's solver time, not the
            # slowest wave's of 256 (csrc/rollout_lunar.hip; bit-identical to the loop below).  Its T forwards read the wide
            # weights from an image packed here, once per rollout (gymrl_mhc_policy_pack: the parameters do not move before the update)
            if POLICY_IMAGE:
                self._pol_image = ops.mhc_policy_pack(desc, getattr(self, "_pol_image", None))
                desc.image = self._pol_image.data_ptr()
            else:
                desc.image = None
            ops.rollout_lunar_mhc(env.state, b.N, env.seed, env.env_id0, c0, b.states, b.actions, b.log_probs, b.values, b.rewards,
                                  b.dones, b.ep_returns, b.next_value, desc, b.T, 0, b.T, cfg.gamma, cfg.lam_actor,
                                  ent=b.old_entropies, lam2=cfg.lam_critic, noise_exp=noise,
                                  gae_running=self._gae_run[0] if fuse_gae else None,
                                  gae_running2=self._gae_run[1] if fuse_gae else None,
                                  gae_workspace=self._gae_ws if fuse_gae else None, ep_stats=env.ep_stats,
                                  refill=getattr(cfg, "rollout_refill", True))
            self.step_count += b.T * b.N
            self.rollout_count += 1
            self._agg_ready = fuse_gae
            return
        for t in range(b.T):
            if desc is not None:
                logits, value = ops.mhc_policy(desc, b.states[t], *self._pol_out)
            elif graphed:                                 # the ~20-launch mHC forward as one graph launch
                logits, value = self._forward_graphed(b.states[t])
            else:
                out = self.model.forward_inference(b.states[t]) if FUSED_INFERENCE else None
                logits, value = self.model(b.states[t]) if out is None else out
            ops.categorical_sample(logits, value=value.view(-1), noise_exp=None if noise is None else noise[t],
                                   seed=env.seed, counter=c0 + t, env_id0=env.env_id0,
                                   act_out=b.actions[t], logp_out=b.log_probs[t], ent_out=b.old_entropies[t],
                                   value_out=b.values[t], online=online(t - 1) if fuse_gae and t > 0 else None)
            env.step(b.actions[t], b.states[t + 1], b.rewards[t], done_out=b.dones[t], ep_ret_out=b.ep_returns[t])
        self.step_count += b.T * b.N
        self.rollout_count += 1
        b.next_value.copy_(self.model.get_value(b.states[b.T]))
        if fuse_gae:
            ops.gae_online_flush(online(b.T - 1), b.next_value)
        self._agg_ready = fuse_gae

    @torch.no_grad()
    def _forward_graphed(self, x):
        """model(x) for the rollout through a captured hipGraph: fixed input / output buffers; the parameters are
        views of the flat buffer the optimiser updates in place, so one capture serves the whole run."""
        if self._fwd_in is None:
            self._fwd_in = torch.empty_like(x)
        self._fwd_in.copy_(x)
        if self._fwd_graph is None:
            if self._fwd_warm < 2:
                self._fwd_warm += 1
                out = self.model.forward_inference(self._fwd_in) if FUSED_INFERENCE else None
                return self.model(self._fwd_in) if out is None else out
            self._fwd_graph = torch.cuda.CUDAGraph()
            with gcapture(self._fwd_graph):
                out = self.model.forward_inference(self._fwd_in) if FUSED_INFERENCE else None
                self._fwd_out = self.model(self._fwd_in) if out is None else out
        self._fwd_graph.replay()
        return self._fwd_out

    def compute_advantages(self):
        """:507-535 -> (adv_actor [T,N] un-normalised, returns [T,N])."""
        b, cfg = self.buffer, self.cfg
        variant = getattr(cfg, "gae_variant", 1)
        if variant == 1 and self._agg_ready:
            variant = 2                                   # the rollout composed both chunk maps
        self._agg_ready = False
        return ops.gae_decoupled(b.rewards, b.values, b.dones, b.next_value, cfg.gamma, cfg.lam_actor, cfg.lam_critic,
                                 variant, self._gae_ws, self._adv, self._ret)

    def update_model(self, advantages, returns):
        """:537-679.  Returns the metric means the reference prints."""
        cfg, b = self.cfg, self.buffer
        total = b.T * b.N
        mb = max(1, total // int(cfg.num_minibatches)) if cfg.num_minibatches else min(int(cfg.batch_size), total)
        n_mb = (total + mb - 1) // mb
        states = b.states[:b.T].reshape(total, -1)
        act, lp, ent_old = b.actions.view(-1), b.log_probs.view(-1), b.old_entropies.view(-1)
        if not torch.is_tensor(advantages):        # the reference hands numpy arrays over (:537-556): one H2D copy each
            advantages = torch.as_tensor(np.asarray(advantages, np.float32), device=self.device)
            returns = torch.as_tensor(np.asarray(returns, np.float32), device=self.device)
        adv, ret = advantages.reshape(-1), returns.reshape(-1)
        metrics = torch.zeros(cfg.num_epochs * n_mb, 9, dtype=torch.float64, device=self.device)
        sizes, row = [], 0
        lcfg = (cfg.clip_eps_min, cfg.clip_eps_max, cfg.dual_clip, cfg.erc_beta_low, cfg.erc_beta_high, self.ent_coef)

        ent_dev = [None]          # graphed: the annealed entropy coefficient read from the device (the graphs outlive this call)
        micro = int(getattr(cfg, "micro_batch", 0) or 0)
        n_micro = mb // micro if (0 < micro < mb and mb % micro == 0 and total % mb == 0) else 1
        rows = mb // n_micro                                               # rows per forward/backward pass
        acc = n_micro > 1

        def fwd_bwd(idx, metrics_row):
            rows_in = (ops.gather_rows(states, idx) if states.shape[1] % 4 == 0 and idx.dtype == torch.int32 and states.is_contiguous()
                       else states.index_select(0, idx))
            logits, values = self.model(rows_in)
            values = values.view(-1)
            mul = None
            if cfg.clip_cov_ratio > 0:                                     # :594-616 (off by default)
                li = idx.long()
                mul = cov_clip_mask(cfg, logits.detach(), act.index_select(0, li), adv.index_select(0, li), self._perm_gen)
            dlogits, dvalues = ops.ppo_full_loss_fwd_bwd(logits, values, act, lp, ent_old, adv, ret, lcfg, idx=idx,
                                                         metrics_sum=metrics_row, corr_mul=mul, entropy_coef_dev=ent_dev[0])
            self._sink.arm(add=acc)               # (the fused layers write their gradients straight into the flat buffer)
            torch.autograd.backward([logits, values], [dlogits, dvalues])
            self._sink.collect(add=acc)           # accumulating: the optimiser step left the buffer zeroed

        def opt_step(bias=None):
            # every pass scaled its loss by 1 / rows: the mean over the minibatch is their average
            self.optimizer.step(grad_scale=1.0 / (self.world_size * n_micro), bias_dev=bias)

        def reduce_grads():
            # one collective over the flat gradient per optimiser step (after the last micro-batch's backward), on the
            # reducer's communication stream so that bench.py can time it apart from the compute stream's kernels
            if self.collective:
                if self._reducer is None:
                    self._reducer = gdist.GradReducer(self.flat_grads)
                self._reducer.timed = self._time_collectives
                self._reducer.launch(0)
                self._reducer.wait()

        def finish(bias=None):
            reduce_grads()
            if self.grad_norms is not None:                                # tests: the norm clip_grad_norm_ would return
                ops.sqnorm(self.flat_grads, self.optimizer._sq, self.optimizer._ws, 1.0 / (self.world_size * n_micro))
                self.grad_norms.append(float(self.optimizer._sq.sqrt().item()))
            opt_step(bias)

        # One pass of the mHC network is ~400 launches of a few microseconds (8 ms at 1024 rows): with equal
        # minibatches the forward/loss/backward body is captured once per update_model() call (rollout tensors and the
        # loss configuration are constants of the capture; the annealed entropy coefficient, Adam's bias block and the
        # minibatch indices are read from device memory) and replayed once per micro-batch, then clip + Adam as a second
        # graph; the gradient all-reduce of a multi-rank run sits between the two, eager, so the 8-GPU configuration runs the
        # same replayed kernels as one GPU.  The first two passes ever run eagerly (library warm-up).  Both graphs are KEPT
        # across update_model() calls as long as every captured address and constant is the same (`key` below): a capture
        # is ~400 launches recorded on the host with the GPU idle, 19 ms of config 5's 970 ms update when repeated per call.
        graphed = (bool(getattr(cfg, "use_graphs", True)) and total % mb == 0 and cfg.num_epochs * n_mb * n_micro > 2
                   and cfg.clip_cov_ratio <= 0 and self.grad_norms is None)      # the covariance clip reads the host
        graph = graph2 = None
        if graphed and (self._g_idx is None or self._g_idx.numel() != rows):
            from .graphs import StepScalars
            self._scalars = StepScalars(self.device)
            self._g_bias, self._g_off = self._scalars.slot(16, torch.float32)
            self._g_ent, self._g_ent_off = self._scalars.slot(4, torch.float32)
            self._g_idx = torch.empty(rows, dtype=torch.int32, device=self.device)
            self._g_row = torch.zeros(9, dtype=torch.float64, device=self.device)
            self._g_graphs = None
        if graphed:
            ent_dev[0] = self._g_ent
            # every HOST constant the captured launches bake in belongs to the key next to the addresses: Adam's betas / eps
            # (load_state_dict() may change them), the clip settings, the gradient scale's inputs, and the module-level
            # FUSED_* switches that choose which kernels a capture records (lr, the bias corrections and the entropy
            # coefficient are read from device memory and may change freely)
            og = self.optimizer.param_groups[0]
            key = (rows, n_micro, self.world_size, lcfg[:5], tuple(t.data_ptr() for t in (states, act, lp, ent_old, adv, ret)),
                   tuple(p.data_ptr() for p in self.model.parameters()), self.flat_grads.data_ptr(), self.optimizer.m.data_ptr(),
                   self.optimizer.v.data_ptr(), tuple(float(b) for b in og["betas"]), float(og["eps"]),
                   self.optimizer.max_grad_norm, self.optimizer.clamp_abs,
                   tuple(sorted((k, bool(v)) for k, v in globals().items() if k.startswith("FUSED_"))))
            kept = getattr(self, "_g_graphs", None)
            if KEEP_UPDATE_GRAPHS and kept is not None and kept[0] == key:
                graph, graph2 = kept[1], kept[2]
            else:
                self._g_graphs = None                                      # (releases the old graphs' memory pool first)
        for _ in range(cfg.num_epochs):
            if self._parity_perms is not None:                             # parity mode: the DataLoader's shuffle order
                perm = torch.as_tensor(next(self._parity_perms), device=self.device).to(torch.int32)
            else:                                                          # shuffle=True: keyed bijection, no sort (:561-563)
                self._perm_draws += 1
                self._perm = ops.permutation(self._perm_seed, self._perm_draws, total, self.device,
                                             out=self._perm if self._perm is not None and self._perm.numel() == total else None)
                perm = self._perm
            for start in range(0, total, mb):
                idx = perm[start:start + mb]
                B = idx.numel()
                if not graphed:
                    for j in range(n_micro):
                        fwd_bwd(idx[j * rows:(j + 1) * rows] if acc else idx, metrics[row])
                    finish()
                else:
                    self._g_row.zero_()
                    self._scalars.set(self._g_off, self.optimizer.next_bias())
                    self._scalars.set(self._g_ent_off, struct.pack("f", self.ent_coef))
                    self._scalars.flush()
                    for j in range(n_micro):
                        self._g_idx.copy_(idx[j * rows:(j + 1) * rows])
                        if self._g_warm < 2:
                            self._g_warm += 1
                            fwd_bwd(self._g_idx, self._g_row)
                        else:
                            if graph is None:                              # (capturing does not execute)
                                graph = torch.cuda.CUDAGraph()
                                with gcapture(graph):
                                    fwd_bwd(self._g_idx, self._g_row)
                            graph.replay()
                    reduce_grads()
                    if graph2 is None:
                        if graph is None:
                            opt_step(self._g_bias)                         # still warming up
                        else:
                            graph2 = torch.cuda.CUDAGraph()
                            with gcapture(graph2):
                                opt_step(self._g_bias)
                            graph2.replay()
                    else:
                        graph2.replay()
                    metrics[row].copy_(self._g_row)
                sizes.append(B)
                row += 1
        if graphed and KEEP_UPDATE_GRAPHS and graph is not None and graph2 is not None:
            self._g_graphs = (key, graph, graph2)
        del graph, graph2
        if cfg.anneal:                                                     # :660-666 (after the update)
            frac = 1 - self.step_count * self.world_size / cfg.max_train_steps
            self.lr = cfg.lr * frac
            for param_group in self.optimizer.param_groups:
                param_group["lr"] = self.lr
            self.ent_coef = cfg.entropy_coef * frac
        done = b.ep_returns[b.dones.bool()][-self.episode_rewards.maxlen:]
        for r in done.tolist():
            self.episode_rewards.append(r)
        m = metrics.cpu().numpy()
        Bs = np.asarray(sizes, np.float64)
        cov = (m[:, 8] - m[:, 6] * m[:, 7] / Bs) / Bs                      # covs.mean() per minibatch (:594-596)
        return {"policy_loss": float((m[:, 0] / Bs).mean()), "value_loss": float((m[:, 1] / Bs).mean()),
                "entropy": float((m[:, 2] / Bs).mean()), "clip_frac": float((m[:, 3] / Bs).mean()),
                "approx_kl": float((m[:, 4] / Bs).mean()), "erc_clip_frac": float((m[:, 5] / Bs).mean()),
                "cov": float(cov.mean())}

    def train(self):
        """:681-700."""
        return self._train()

    def _train(self):
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

    @torch.no_grad()
    def eval(self, num_episodes=10):
        env = VecEnv(self.cfg.env_name, num_episodes, device=self.device, seed=self.base_seed + 1_000_003,
                     env_id0=1 << 40)
        obs = env.reset()
        nxt = torch.empty_like(obs)
        rew = torch.empty(num_episodes, device=self.device)
        done = torch.zeros(num_episodes, dtype=torch.uint8, device=self.device)
        ep_ret = torch.zeros(num_episodes, device=self.device)
        result = torch.full((num_episodes,), float("nan"), device=self.device)
        for _ in range(env.max_steps + 1):
            act = self.model.get_action(obs, deterministic=True)[0]
            env.step(act, nxt, rew, done_out=done, ep_ret_out=ep_ret)
            result = torch.where(done.bool() & torch.isnan(result), ep_ret, result)
            obs, nxt = nxt, obs
            if not torch.isnan(result).any():
                break
        return result.tolist()

    def test(self):
        return self.eval(num_episodes=5)


if __name__ == "__main__":       # python -m gymrl_amd.ppo_full_lunarlander [--<Config attribute> <value> ...]  (ppo_full_lunarlander.py:750-767)
    from .utils.cli import run_script
    run_script(Config, PPOTrainer, interrupted="\nCtrl+C detected, stopping training and starting test...")
