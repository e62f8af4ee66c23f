"""Linear layer whose weight gradient is a batched split-K GEMM.

The PPO update runs minibatches of 10^5..10^6 rows through 256-wide layers, so the weight
gradient dW = dY^T X is a skinny-output GEMM with a huge reduction dimension (K = batch).
hipBLASLt's pick for that shape reaches 42 TF/s on MI355X (806 us at 262144 x 256 x 256);
reshaping the batch into S independent slices turns it into a batched GEMM that fills all
256 CUs (275 us, 125 TF/s — `tools/micro_linear.py`), followed by a 16 MB sum over slices.
Same parameters, same state_dict keys as nn.Linear; small batches take the stock path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_MIN_ROWS = 16384


class _SplitKLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        B, N = dy.shape
        K = x.shape[1]
        S = 256 if N == 1 else 128
        while B % S:
            S //= 2
        dx = dy.mm(weight) if ctx.needs_input_grad[0] else None
        dw = torch.bmm(dy.view(S, B // S, N).transpose(1, 2), x.view(S, B // S, K)).sum(0)
        return dx, dw, dy.sum(0)


class Linear(nn.Linear):
    def forward(self, x):
        if (x.dim() == 2 and x.shape[0] >= _MIN_ROWS and x.is_contiguous() and torch.is_grad_enabled()
                and self.weight.requires_grad and self.bias is not None):
            return _SplitKLinear.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


SPLIT_BIAS = True      # tools/micro_offpolicy.py flips this for its A/B


def small_linear(x, weight, bias):
    """F.linear for the 128-8192 row batches of the off-policy networks as a plain GEMM + in-place bias add.
    torch.addmm's fused-bias form always goes to hipBLASLt, whose default pick for these shapes is one
    256 x 256 macro tile (37 us at 256 or 8192 rows x 256 x 256, rocprofv3: 6 such calls = a quarter of a Rainbow
    vector step); the plain GEMM follows gymrl_amd/blas.py's library preference (rocBLAS: 6-15 us) and the bias
    costs one 4 us elementwise launch."""
    if SPLIT_BIAS and x.dim() == 2 and bias is not None:
        return torch.mm(x, weight.t()).add_(bias)
    return F.linear(x, weight, bias)


class SmallLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) with `small_linear` as its forward."""

    def forward(self, x):
        return small_linear(x, self.weight, self.bias)


class FusedMLP:
    """Inference forward of a Linear(+Tanh|ReLU) network as ONE launch (`gymrl_mlp_forward`,
    csrc/mlp.hip: 16 rows per workgroup, activations in LDS, f32 MFMA).  Built from a list of
    (linear_module, activation, src, dst) stages.  The kernel reads a packed copy of each weight
    (the MFMA B-operand image, `gymrl_mlp_pack`): call refresh() after the parameters changed —
    once per rollout in the PPO trainer — biases are read in place.  Outputs of dst == -1 stages
    are allocated once per batch size and reused.

        fused = FusedMLP([(net.shared[0], "tanh", -1, 0), (net.shared[2], "tanh", 0, 1),
                          (net.actor[0], "tanh", 1, 0), (net.actor[2], None, 0, -1),
                          (net.critic[0], "tanh", 1, 0), (net.critic[2], None, 0, -1)])
        logits, value = fused(obs)
    """

    _ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}

    def __init__(self, stages):
        self.stages = list(stages)
        self._cache = {}
        self._packed = None

    def refresh(self):
        """Re-pack every layer's weight (one tiny launch per layer)."""
        from . import ops
        if self._packed is None:
            self._packed = [None] * len(self.stages)
        for i, (m, _, _, _) in enumerate(self.stages):
            self._packed[i] = ops.mlp_pack(m.weight.detach(), self._packed[i])

    @staticmethod
    def supported(stages, in_dim):
        from . import ops
        if in_dim > ops.MLP_MAX_INPUT or len(stages) > ops.MLP_MAX_STAGES:
            return False
        return all(m.in_features <= ops.MLP_MAX_WIDTH and (dst < 0 or m.out_features <= ops.MLP_MAX_WIDTH)
                   for m, _, _, dst in stages)

    def _build(self, n, device):
        from . import ops
        outs, table = [], []
        for (m, act, src, dst), packed in zip(self.stages, self._packed):
            out = torch.empty(n, m.out_features, device=device) if dst < 0 else None
            if out is not None:
                outs.append(out)
            table.append(dict(W=packed, shape=(m.out_features, m.in_features),
                              b=None if m.bias is None else m.bias.detach(),
                              act=self._ACT[act], src=src, dst=dst, out=out))
        ptrs = tuple(m.bias.data_ptr() for m, _, _, _ in self.stages if m.bias is not None)
        return ops.mlp_desc(table), outs, ptrs

    def descriptor(self, n, device):
        """The gymrl_mlp_desc of this network for n rows (packs the weights on first use)."""
        if self._packed is None:
            self.refresh()
        key = (n, device)
        ent = self._cache.get(key)
        if ent is None or ent[2] != tuple(m.bias.data_ptr() for m, _, _, _ in self.stages if m.bias is not None):
            ent = self._cache[key] = self._build(n, device)
        return ent[0]

    @torch.no_grad()
    def __call__(self, x, refresh=False):
        from . import ops
        if refresh or self._packed is None:
            self.refresh()
        key = (x.shape[0], x.device)
        ent = self._cache.get(key)
        if ent is None or ent[2] != tuple(m.bias.data_ptr() for m, _, _, _ in self.stages if m.bias is not None):
            ent = self._cache[key] = self._build(x.shape[0], x.device)
        ops.mlp_forward(x, ent[0])
        return ent[1]
