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
    vector step); the plain GEMM is the library's (rocBLAS answers these shapes in 6-15 us; tools/blas_pref.py scopes the preference for A/B runs) and the bias
    costs one 4 us elementwise launch."""
    if SPLIT_BIAS and x.dim() == 2 and bias is not None:
        return torch.mm(x, weight.t()).add_(bias)
    return F.linear(x, weight, bias)


FUSED_LINEAR = True    # tools/micro_offpolicy.py flips this for its A/B (False: library GEMM + elementwise launches)
_FUSED_MAX_ROWS = 16384


class _FusedLinear(torch.autograd.Function):
    """n Linear(+activation) layers of one shape as ONE launch per direction (csrc/lin.hip): forward
    y_i = act_i(cat(x_i, x2_i) W_i^T + b_i); backward one launch for every input gradient and one for every weight +
    bias gradient, the activation's derivative taken from the saved outputs inside those launches."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        from . import ops
        n, has_x2, acts, los, his = spec
        xs = list(tensors[:n])
        x2s = list(tensors[n:2 * n]) if has_x2 else [None] * n
        rest = tensors[(2 if has_x2 else 1) * n:]
        ws, bs = list(rest[:n]), list(rest[n:2 * n])
        # one input feeding every item (two heads on a trunk, the twin critics' first layers): its gradient is the sum
        ctx.shared = n > 1 and all(x is xs[0] for x in xs) and all(x is x2s[0] for x in x2s)
        x0 = xs[0].contiguous()
        xs = [x0] * n if ctx.shared else [x.contiguous() for x in xs]
        if has_x2:
            x20 = x2s[0].contiguous()
            x2s = [x20] * n if ctx.shared else [x.contiguous() for x in x2s]
        ys = ops.lin_fwd(xs, ws, bs, list(acts), x2=x2s, lo=list(los), hi=list(his))
        ctx.spec = spec
        ctx.save_for_backward(*xs, *[x for x in x2s if x is not None], *ws, *ys)
        ctx.has_bias = [b is not None for b in bs]
        ctx.sinks = [(getattr(w, "_gymrl_sink", None), None if b is None else getattr(b, "_gymrl_sink", None))
                     for w, b in zip(ws, bs)]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        from . import ops
        n, has_x2, acts, los, his = ctx.spec
        saved = ctx.saved_tensors
        xs = list(saved[:n])
        x2s = list(saved[n:2 * n]) if has_x2 else [None] * n
        o = (2 if has_x2 else 1) * n
        ws, ys = list(saved[o:o + n]), list(saved[o + n:o + 2 * n])
        dys = [torch.zeros_like(y) if d is None else d.contiguous() for d, y in zip(dys, ys)]
        need = ctx.needs_input_grad[1:]
        need_x, need_x2 = any(need[:n]), has_x2 and any(need[n:2 * n])
        need_w = any(need[o:o + n])
        dxs, dx2s = [None] * n, [None] * n
        if need_x or need_x2:
            got = ops.lin_bwd_input(dys, ys, ws, list(acts), K1=xs[0].shape[1], want=(need_x, need_x2),
                                    lo=list(los), hi=list(his), sum_items=ctx.shared)
            if ctx.shared:
                dxs[0], dx2s[0] = got
            else:
                dxs, dx2s = got
        dws, dbs = [None] * n, [None] * n
        if need_w:
            # straight into the flat gradient buffer where the parameters' GradSink is armed (no fresh tensors, no
            # gather launch afterwards); fresh tensors handed to autograd otherwise
            slots = [GradSink_direct(sw, sb, hb) for (sw, sb), hb in zip(ctx.sinks, ctx.has_bias)]
            if all(s is not None for s in slots) and len({s[2] for s in slots}) == 1:
                ops.lin_bwd_weight(dys, ys, xs, [s[0] for s in slots], [s[1] for s in slots], list(acts), x2=x2s,
                                   lo=list(los), hi=list(his), accumulate=slots[0][2])
            else:
                for s, (sw, sb) in zip(slots, ctx.sinks):
                    if s is not None:
                        GradSink_undo(sw, sb)
                dws = [torch.empty_like(w) for w in ws]
                dbs = [torch.empty(w.shape[0], dtype=w.dtype, device=w.device) if hb else None
                       for w, hb in zip(ws, ctx.has_bias)]
                ops.lin_bwd_weight(dys, ys, xs, dws, dbs, list(acts), x2=x2s, lo=list(los), hi=list(his))
        return (None, *dxs, *(dx2s if has_x2 else ()), *dws, *dbs)


def GradSink_direct(sw, sb, has_bias):
    """(weight view, bias view | None, accumulate) of a layer whose parameters belong to an armed GradSink
    (gymrl_amd/flat.py), else None."""
    if sw is None or (has_bias and sb is None) or not sw[0].armed or (has_bias and sb[0] is not sw[0]):
        return None
    sink = sw[0]
    vw, aw = sink.direct(sw[1])
    if not has_bias:
        return vw, None, aw
    vb, ab = sink.direct(sb[1])
    if ab != aw:
        sink.undo(sw[1]), sink.undo(sb[1])
        return None
    return vw, vb, aw


def GradSink_undo(sw, sb):
    sw[0].undo(sw[1])
    if sb is not None:
        sb[0].undo(sb[1])


def _fusable(x, weight):
    """Batches up to a few thousand rows — and skinny layers (<= 16 inputs or outputs) at ANY row count: the library answers
    a [262144, 256] x [256, 8] product with a 256 x 16 macro tile in 3.6 ms (PPO-full's gate read-outs and heads: 78 % of its
    update, `profiles/r02_ppo_full_kernel_stats.csv`), where the operands' 268 MB are 60 us of HBM time."""
    return (FUSED_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.shape[0] > 0 and (x.shape[0] <= _FUSED_MAX_ROWS or min(weight.shape) <= 16))


def _wide(x, weight):
    """Layers too wide for the layer kernels at >= 16384 rows whose weight gradient gymrl_lin_bwd_weight's 64 x 64-block
    kernel takes (rows >= 16384, both widths multiples of 64)."""
    return (FUSED_LINEAR and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.shape[0] >= 16384 and weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0)


class _WideLinear(torch.autograd.Function):
    """A Linear layer beyond `_fusable` at a large batch (PPO-full's 128 -> 256 head layers at 262144-row micro-batches):
    forward and input gradient are gymrl_linear_fwd / gymrl_linear_bwd_input (csrc/gemm.hip: weight-stationary exact-f32
    MFMA kernels; the 64- / 128-long reductions stage their row tiles through LDS) for the shapes they cover
    (`ops.linear_shape_ok`), the weight + bias gradient one gymrl_lin_bwd_weight call (64 x 64 blocks per wave: 110 us
    where the library's [N, B] x [B, K] product takes 491 plus a column-sum launch), written straight into an armed
    GradSink's buffer.  Other shapes: the library GEMM."""

    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.sinks = (getattr(w, "_gymrl_sink", None), None if b is None else getattr(b, "_gymrl_sink", None))
        ctx.own = ops.linear_shape_ok(w.shape[1], w.shape[0]) and w.is_contiguous() and x.data_ptr() % 16 == 0
        if ctx.own:
            return ops.linear_fwd(x, w, b, torch.empty(x.shape[0], w.shape[0], device=x.device), act=False)
        return F.linear(x, w, b)                     # bias in the GEMM's epilogue: a separate add is a 270 MB pass at 262144 rows

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.own and dy.data_ptr() % 16 == 0:
                dx = ops.linear_bwd_input(dy, w, None, torch.empty_like(x))
            else:
                dx = torch.mm(dy, w)
        dw = db = None
        if ctx.needs_input_grad[1]:
            slot = GradSink_direct(ctx.sinks[0], ctx.sinks[1], ctx.has_bias)
            if slot is not None:
                ops.lin_bwd_weight(dy, None, x, slot[0], slot[1], accumulate=slot[2])
            else:
                dw = torch.empty_like(w)
                db = torch.empty(w.shape[0], dtype=w.dtype, device=w.device) if ctx.has_bias else None
                ops.lin_bwd_weight(dy, None, x, dw, db)
        return dx, dw, db


class _WideLinearPair(torch.autograd.Function):
    """Two `_WideLinear` layers of the (256, 128) shape on ONE input (PPO-full's actor.mlp.0 and critic.mlp.0 on the backbone's
    output): the same forward launches, and the backward adds the second layer's input gradient to the first one's inside its
    GEMM's epilogue (gymrl_linear_bwd_input_add) — as two autograd nodes the sum was a torch add over [B, 128] per micro-batch,
    16 ms of config 5's update.  a + b either way: the same bits."""

    @staticmethod
    def forward(ctx, x, wa, ba, wc, bc):
        from . import ops
        x = x.contiguous()
        ctx.save_for_backward(x, wa, wc)
        ctx.sinks = [(getattr(w, "_gymrl_sink", None), getattr(b, "_gymrl_sink", None)) for w, b in ((wa, ba), (wc, bc))]
        B = x.shape[0]
        ya = ops.linear_fwd(x, wa, ba, torch.empty(B, wa.shape[0], device=x.device), act=False)
        yc = ops.linear_fwd(x, wc, bc, torch.empty(B, wc.shape[0], device=x.device), act=False)
        return ya, yc

    @staticmethod
    def backward(ctx, dya, dyc):
        from . import ops
        x, wa, wc = ctx.saved_tensors
        dya, dyc = dya.contiguous(), dyc.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            ga = ops.linear_bwd_input(dya, wa, None, torch.empty_like(x))
            dx = ops.linear_bwd_input_add(dyc, wc, ga, torch.empty_like(x))
        grads = []
        for k, (dy, w) in enumerate(((dya, wa), (dyc, wc))):
            dw = db = None
            if ctx.needs_input_grad[1 + 2 * k]:
                slot = GradSink_direct(ctx.sinks[k][0], ctx.sinks[k][1], True)
                if slot is not None:
                    ops.lin_bwd_weight(dy, None, x, slot[0], slot[1], accumulate=slot[2])
                else:
                    dw, db = torch.empty_like(w), torch.empty(w.shape[0], dtype=w.dtype, device=w.device)
                    ops.lin_bwd_weight(dy, None, x, dw, db)
            grads += [dw, db]
        return (dx, *grads)


class _SmallKLinear(torch.autograd.Function):
    """A Linear layer on 2 / 3 / 4 / 8 inputs at a large batch (PPO-full's input projection at 524 288-row micro-batches): the
    forward is gymrl_linear_smallk (one fmaf chain per output, 16-byte stores: HBM-bound), the weight + bias gradient the layer
    kernels' (gymrl_lin_bwd_weight, into an armed GradSink's buffer).  No input gradient: the input is the observation."""

    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.sinks = (getattr(w, "_gymrl_sink", None), getattr(b, "_gymrl_sink", None))
        return ops.linear_smallk(x, w, b, torch.empty(x.shape[0], w.shape[0], device=x.device))

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.mm(dy, w) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            slot = GradSink_direct(ctx.sinks[0], ctx.sinks[1], True)
            if slot is not None:
                ops.lin_bwd_weight(dy, None, x, slot[0], slot[1], accumulate=slot[2])
            else:
                dw, db = torch.empty_like(w), torch.empty(w.shape[0], dtype=w.dtype, device=w.device)
                ops.lin_bwd_weight(dy, None, x, dw, db)
        return dx, dw, db


def smallk_linear(x, layer):
    """layer(x) through `_SmallKLinear` where it applies (>= 16384 rows, 2 / 3 / 4 / 8 inputs, a power-of-two width, bias, no
    activation), else None."""
    w = layer.weight
    ok = (FUSED_LINEAR and isinstance(layer, SmallLinear) and layer.act in (None, "none") and layer.bias is not None
          and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 16384 and w.shape[1] in (2, 3, 4, 8)
          and w.shape[0] >= 4 and (w.shape[0] & (w.shape[0] - 1)) == 0 and w.is_contiguous() and x.is_contiguous()
          and x.data_ptr() % 16 == 0)
    if not ok:
        return None
    return _SmallKLinear.apply(x, w, layer.bias)


def wide_linear_pair(x, la, lc):
    """(la(x), lc(x)) for two bias-carrying (256, 128) `SmallLinear` layers without activation at >= 16384 rows, or None when
    the pair is not that shape (the caller then runs the layers one by one)."""
    from . import ops
    ok = (all(isinstance(l, SmallLinear) and l.act in (None, "none") and l.bias is not None and tuple(l.weight.shape) == (256, 128)
              and l.weight.is_contiguous() and _wide(x, l.weight) and not _fusable(x, l.weight) for l in (la, lc))
          and x.is_contiguous() and x.data_ptr() % 16 == 0 and ops.linear_shape_ok(128, 256))
    return _WideLinearPair.apply(x, la.weight, la.bias, lc.weight, lc.bias) if ok else None


def _act_torch(z, act, clamp):
    if act == "relu":
        return F.relu(z)
    if act == "tanh":
        return torch.tanh(z)
    if act == "clamp":
        return z.clamp(clamp[0], clamp[1])
    return z


def fused_linears(layers, xs, x2s=None):
    """[layer_i(cat(x_i, x2_i))] for `SmallLinear` layers of one shape in one launch per direction — the twin Q
    networks' parallel layers, two heads on one trunk (pass the same x twice).  x2s: second input blocks (the critic's
    action columns) so that torch.cat never materialises."""
    from .ops import LIN_ACT
    n = len(layers)
    if not _fusable(xs[0], layers[0].weight):
        outs = []
        for i, (layer, x) in enumerate(zip(layers, xs)):
            xin = x if x2s is None else torch.cat([x, x2s[i]], dim=1)
            if _wide(xin, layer.weight):
                z = _WideLinear.apply(xin, layer.weight, layer.bias)
            else:
                z = small_linear(xin, layer.weight, layer.bias)
            outs.append(_act_torch(z, layer.act, layer.clamp))
        return outs
    spec = (n, x2s is not None, tuple(LIN_ACT[l.act] for l in layers), tuple(float(l.clamp[0]) for l in layers),
            tuple(float(l.clamp[1]) for l in layers))
    args = list(xs) + (list(x2s) if x2s is not None else []) + [l.weight for l in layers] + [l.bias for l in layers]
    return list(_FusedLinear.apply(spec, *args))


class SmallLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) for the 64-8192 row batches of the off-policy networks, with
    an optional fused activation: act in (None, "relu", "tanh", "clamp"); clamp = (lo, hi).  On the GPU the layer is one
    launch per direction (`_FusedLinear`, csrc/lin.hip); `FUSED_LINEAR = False` (the A/B in tools/micro_offpolicy.py) and
    CPU tensors (module construction, state_dict round trips) take the library path `small_linear` + activation."""

    def __init__(self, in_features, out_features, bias=True, act=None, clamp=(0.0, 0.0)):
        super().__init__(in_features, out_features, bias=bias)
        self.act, self.clamp = act, clamp

    def forward(self, x, x2=None):
        return fused_linears([self], [x], None if x2 is None else [x2])[0]


class _SkinnyMatmul(torch.autograd.Function):
    """x [B, K] @ w [K, G] with G <= 16 (the mHC gates' read-out, ppo_full_lunarlander.py:129) on the layer kernels:
    forward, input gradient and weight gradient are one HBM-bound launch each at any B."""

    @staticmethod
    def forward(ctx, x, w):
        from . import ops
        x = x.contiguous()
        wt = w.t().contiguous()                      # [G, K] = the nn.Linear layout the kernels take
        ctx.save_for_backward(x, wt)
        return ops.lin_fwd(x, wt, None)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, wt = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.lin_bwd_input(dy, None, wt)[0] if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dwt = torch.empty_like(wt)
            ops.lin_bwd_weight(dy, None, x, dwt, None)
            dw = dwt.t()
        return dx, dw


def skinny_matmul(x, w):
    """x @ w for a narrow w ([K, G], G <= 16) — fused kernels on the GPU, torch elsewhere."""
    if FUSED_LINEAR and x.is_cuda and x.dim() == 2 and w.shape[1] <= 16 and x.dtype == torch.float32:
        return _SkinnyMatmul.apply(x, w)
    return x @ w


class frozen_parameters:
    """Context: the module's parameters do not require gradients inside (the critic during SAC's / TD3's actor step:
    sac_pendulum.py:248-255 computes and then discards them; here the weight-gradient launches are never issued)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


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
