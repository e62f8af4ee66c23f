"""Flat parameter / gradient buffers + the fused Adam front-end.

The reference keeps an nn.Module and a torch.optim.Adam per network
(ppo_lunarlander.py:165-169).  Here every network keeps its nn.Module interface
(so the GEMMs run through PyTorch-ROCm autograd), but all parameters are views
into ONE flat fp32 buffer and all gradients views into another, so that
clip_grad_norm_ + Adam (+ zero_grad) is two kernel launches on contiguous memory
and a multi-GPU gradient all-reduce is one collective on one buffer.
"""
import torch

from . import ops

_ALIGN = 64  # floats (256 B): keeps every parameter view aligned for hipBLASLt and float4 kernels


def flatten_module(module, device, order=None):
    """Move `module` to `device` with all parameters aliased into one flat buffer.
    Returns (flat_params, flat_grads).  `order` (optional list of parameter names) fixes the
    layout of the named parameters at the front of the buffer, in that order — used to make two
    layers' weights adjacent so that they can be driven as one GEMM; state_dict keys are untouched."""
    named = dict(module.named_parameters())
    params = [named[n] for n in (order or [])]
    seen = {id(p) for p in params}
    params += [p for p in module.parameters() if id(p) not in seen]
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    flat = torch.zeros(total, dtype=torch.float32, device=device)
    grad = torch.zeros(total, dtype=torch.float32, device=device)
    module.to(device)
    for p, o in zip(params, offs):
        n = p.numel()
        flat[o:o + n].copy_(p.data.reshape(-1))
        p.data = flat[o:o + n].view(p.shape)
        p.grad = grad[o:o + n].view(p.shape)
    module._flat_params, module._flat_grads = flat, grad
    return flat, grad


class GradSink:
    """Lets autograd hand back FRESH gradient tensors and lands them in the flat buffer with one multi-tensor
    copy.  With `p.grad` pre-set to a view of the flat buffer, autograd accumulates (`grad += new`, one launch
    per parameter per backward: 34 launches in one SAC update); with `p.grad = None` it just stores the new
    tensor.  arm() before backward, collect() after it (or drop() when the gradients are not wanted)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.views = [p.grad for p in self.params]
        self.armed, self.add, self.written = False, False, {}
        for i, p in enumerate(self.params):
            p._gymrl_sink = (self, i)        # the fused layers (gymrl_amd/nn.py) write their gradients straight into the view

    def arm(self, add=False):
        """add=True: this backward accumulates on top of the flat buffer (pass the same flag to collect())."""
        for p in self.params:
            p.grad = None
        self.armed, self.add, self.written = True, add, {}

    def direct(self, i):
        """A fused layer's backward asks where parameter i's gradient goes: (view, accumulate).  The first write of a
        backward overwrites (the buffer may hold anything), later ones — the layer used twice — accumulate."""
        k = self.written.get(i, 0)
        self.written[i] = k + 1
        return self.views[i], bool(self.add or k)

    def undo(self, i):
        k = self.written.get(i, 0) - 1
        if k > 0:
            self.written[i] = k
        else:
            self.written.pop(i, None)

    def collect(self, add=False):
        """add=True accumulates into the flat buffer (gradient accumulation over micro-batches: the buffer was zeroed
        by the optimiser step) instead of overwriting it."""
        if add != self.add and self.written:
            raise RuntimeError("GradSink: arm(add=...) and collect(add=...) disagree while gradients were written in place")
        got = [(v, p.grad, i) for i, (v, p) in enumerate(zip(self.views, self.params)) if p.grad is not None]
        if not add:
            for i, (v, p) in enumerate(zip(self.views, self.params)):
                if p.grad is None and i not in self.written:
                    v.zero_()
        if got:
            first = [(v, g) for v, g, i in got if not (add or i in self.written)]
            more = [(v, g) for v, g, i in got if add or i in self.written]
            if first:
                torch._foreach_copy_([v for v, _ in first], [g for _, g in first])
            if more:
                torch._foreach_add_([v for v, _ in more], [g for _, g in more])
        self.drop()

    def drop(self):
        for v, p in zip(self.views, self.params):
            p.grad = v
        self.armed, self.written = False, {}


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps) + clip_grad_norm_ + zero_grad over one flat
    buffer, through gymrl_sqnorm / gymrl_adam_step.  Keeps the `param_groups[i]["lr"]`
    interface the reference's LR annealing writes to (ppo_lunarlander.py:337-341)."""

    def __init__(self, flat_params, flat_grads, lr, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0,
                 clamp_abs=0.0, module=None):
        self.p, self.g = flat_params, flat_grads
        self.module = module       # when given, state_dict() / load_state_dict() speak torch.optim.Adam's layout
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps)]
        self.max_grad_norm, self.clamp_abs = float(max_grad_norm), float(clamp_abs)
        self.step_count = 0
        self._sq = torch.zeros(1, dtype=torch.float64, device=flat_params.device)
        self._ws = ops.reduce_workspace(flat_params.device)

    def zero_grad(self):
        """No-op: gymrl_adam_step zeroes the gradient buffer it just consumed."""

    def step(self, grad_scale=1.0, bias_dev=None, polyak=None):
        """bias_dev (f32[4] device view, hipGraph replay): the step-dependent scalars come from the device and
        the caller advances the step with next_bias().  polyak = (target_flat, tau): the soft target update of the
        freshly written parameters in the same launch."""
        g = self.param_groups[0]
        if bias_dev is None:
            self.step_count += 1
        if self.max_grad_norm > 0:      # the norm's second level inside the Adam launch: two launches, not three
            ops.clip_adam_step(self.p, self.g, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                               max(self.step_count, 1), self.max_grad_norm, self._ws, grad_scale=grad_scale, sqnorm_out=self._sq,
                               clamp_abs=self.clamp_abs, zero_grad=True, bias_dev=bias_dev,
                               polyak_target=None if polyak is None else polyak[0], tau=0.0 if polyak is None else polyak[1])
            return
        ops.adam_step(self.p, self.g, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                      max(self.step_count, 1), grad_scale=grad_scale, max_grad_norm=self.max_grad_norm,
                      sqnorm_buf=self._sq, clamp_abs=self.clamp_abs, zero_grad=True, bias_dev=bias_dev,
                      polyak_target=None if polyak is None else polyak[0], tau=0.0 if polyak is None else polyak[1])

    def next_bias(self):
        """Advance the step count and return the 16-byte payload of gymrl_adam_bias for it."""
        g = self.param_groups[0]
        self.step_count += 1
        return ops.adam_bias(g["lr"], g["betas"][0], g["betas"][1], self.step_count)

    def state_dict(self):
        if self.module is not None:
            from .utils.checkpoint import adam_state_dict
            return adam_state_dict(self.module, self)
        return dict(m=self.m, v=self.v, step=self.step_count, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        if "state" in sd and self.module is not None:
            from .utils.checkpoint import load_adam_state_dict
            return load_adam_state_dict(self.module, self, sd)
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        self.step_count = int(sd["step"])
        self.param_groups[0].update(sd["param_groups"][0])
