"""Low-latency Linear layers (csrc/lin.hip: gymrl_lin_fwd / _bwd_input / _bwd_weight) against torch float64.

Floating-point kernels: the accumulation order differs from any library GEMM, so the comparison is against the
float64 result of the same expression with a relative tolerance of 1e-5 of the output scale (f32 sums of <= 8192
products of O(1) terms); exact zeros / masks must be exact.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

ACTS = {"none": 0, "tanh": 1, "relu": 2, "clamp": 3}
LO, HI = -0.7, 0.9


def _act64(z, act):
    if act == "relu":
        return torch.relu(z)
    if act == "tanh":
        return torch.tanh(z)
    if act == "clamp":
        return z.clamp(LO, HI)
    return z


def _dact64(y, act):
    if act == "relu":
        return (y > 0).double()
    if act == "tanh":
        return 1 - y * y
    if act == "clamp":       # the mask is taken on the float32 outputs with float32 bounds
        y32 = y.float()
        return ((y32 > torch.tensor(LO, dtype=torch.float32)) & (y32 < torch.tensor(HI, dtype=torch.float32))).double()
    return torch.ones_like(y)


def _close(got, ref, scale=None, tol=1e-5):
    ref = ref.to(got.device)
    s = float(ref.abs().max()) if scale is None else scale
    err = float((got.double() - ref).abs().max())
    assert err <= tol * max(s, 1e-3), (err, s)


SHAPES = [  # B, K1, K2, N
    (128, 256, 0, 256), (128, 3, 1, 256), (128, 256, 0, 1), (256, 4, 0, 128), (100, 8, 0, 51), (37, 130, 5, 70),
    (4096, 256, 0, 256), (1, 3, 0, 256), (4096, 3, 0, 256), (8192, 128, 0, 2),
    (20000, 256, 0, 8), (20000, 8, 0, 128), (16385, 12, 0, 4),          # skinny shapes at large B: the VALU kernels
]


@pytest.mark.parametrize("B,K1,K2,N", SHAPES)
@pytest.mark.parametrize("act", ["none", "relu", "tanh", "clamp"])
def test_lin_fwd(B, K1, K2, N, act):
    from gymrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 7 + N + K1)
    d = "cuda"
    K = K1 + K2
    for n_items in (1, 2):
        xs = [torch.randn(B, K1, generator=g).to(d) for _ in range(n_items)]
        x2s = [torch.randn(B, K2, generator=g).to(d) if K2 else None for _ in range(n_items)]
        ws = [(torch.randn(N, K, generator=g) / K ** 0.5).to(d) for _ in range(n_items)]
        bs = [torch.randn(N, generator=g).to(d) for _ in range(n_items)]
        if n_items == 1:
            ys = [ops.lin_fwd(xs[0], ws[0], bs[0], ACTS[act], x2=x2s[0], lo=LO, hi=HI)]
        else:
            ys = ops.lin_fwd(xs, ws, bs, ACTS[act], x2=x2s, lo=LO, hi=HI)
        for x, x2, w, b, y in zip(xs, x2s, ws, bs, ys):
            xc = x.double() if x2 is None else torch.cat([x, x2], 1).double()
            z = xc @ w.double().t() + b.double()
            _close(y, _act64(z, act), scale=float(z.abs().max()), tol=2e-6 if act != "tanh" else 1e-5)
    # no bias, output into a column slice of a wider buffer
    big = torch.full((B, N + 5), 7.0, device=d)
    ops.lin_fwd(xs[0], ws[0], None, ACTS[act], x2=x2s[0], out=big[:, 2:2 + N], lo=LO, hi=HI)
    xc = xs[0].double() if x2s[0] is None else torch.cat([xs[0], x2s[0]], 1).double()
    z = xc @ ws[0].double().t()
    _close(big[:, 2:2 + N], _act64(z, act), scale=float(z.abs().max()), tol=2e-6 if act != "tanh" else 1e-5)
    assert bool((big[:, :2] == 7.0).all()) and bool((big[:, 2 + N:] == 7.0).all())


@pytest.mark.parametrize("B,K1,K2,N", SHAPES)
@pytest.mark.parametrize("act", ["none", "relu", "tanh", "clamp"])
def test_lin_bwd_input(B, K1, K2, N, act):
    from gymrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 11 + N + K1)
    d = "cuda"
    K = K1 + K2
    for n_items in (1, 2):
        dys = [torch.randn(B, N, generator=g).to(d) for _ in range(n_items)]
        ys = [_act64(torch.randn(B, N, generator=g).double(), act).float().to(d) for _ in range(n_items)]
        ws = [(torch.randn(N, K, generator=g) / N ** 0.5).to(d) for _ in range(n_items)]
        if n_items == 1:
            dx, dx2 = ops.lin_bwd_input(dys[0], ys[0], ws[0], ACTS[act], K1=K1, lo=LO, hi=HI)
            dx, dx2 = [dx], [dx2]
        else:
            dx, dx2 = ops.lin_bwd_input(dys, ys, ws, ACTS[act], K1=K1, lo=LO, hi=HI)
        for i in range(n_items):
            ref = (dys[i].double() * _dact64(ys[i].double(), act)) @ ws[i].double()
            _close(dx[i], ref[:, :K1], scale=float(ref.abs().max()), tol=2e-6)
            if K2:
                _close(dx2[i], ref[:, K1:], scale=float(ref.abs().max()), tol=2e-6)
            else:
                assert dx2[i] is None
    # only the second block, accumulated into an existing buffer
    if K2:
        acc = torch.ones(B, K2, device=d)
        a, b = ops.lin_bwd_input(dys[0], ys[0], ws[0], ACTS[act], K1=K1, dx2=acc, want=(False, True), lo=LO, hi=HI,
                                 accumulate=True)
        assert a is None and b is acc
        ref = (dys[0].double() * _dact64(ys[0].double(), act)) @ ws[0].double()
        _close(acc, ref[:, K1:] + 1.0, scale=float(ref.abs().max()) + 1.0, tol=2e-6)


BIG = [(40013, 128, 0, 128), (16384, 128, 0, 256), (33000, 64, 0, 192)]   # >= 16384 rows, widths % 64: the 64 x 64-block kernel


@pytest.mark.parametrize("B,K1,K2,N", SHAPES + [(1024, 256, 0, 256), (700, 4, 2, 33)] + BIG)
@pytest.mark.parametrize("act", ["none", "relu", "tanh"])
def test_lin_bwd_weight(B, K1, K2, N, act):
    from gymrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 13 + N + K1)
    d = "cuda"
    K = K1 + K2
    for n_items in (1, 2):
        dys = [torch.randn(B, N, generator=g).to(d) for _ in range(n_items)]
        ys = [_act64(torch.randn(B, N, generator=g).double(), act).float().to(d) for _ in range(n_items)]
        xs = [torch.randn(B, K1, generator=g).to(d) for _ in range(n_items)]
        x2s = [torch.randn(B, K2, generator=g).to(d) if K2 else None for _ in range(n_items)]
        dws = [torch.full((N, K), 3.0, device=d) for _ in range(n_items)]
        dbs = [torch.full((N,), 3.0, device=d) for _ in range(n_items)]
        if n_items == 1:
            ops.lin_bwd_weight(dys[0], ys[0], xs[0], dws[0], dbs[0], ACTS[act], x2=x2s[0])
        else:
            ops.lin_bwd_weight(dys, ys, xs, dws, dbs, ACTS[act], x2=x2s)
        for i in range(n_items):
            dz = dys[i].double() * _dact64(ys[i].double(), act)
            xc = xs[i].double() if x2s[i] is None else torch.cat([xs[i], x2s[i]], 1).double()
            scale = (B ** 0.5) * 3
            _close(dws[i], dz.t() @ xc, scale=scale, tol=2e-6)
            _close(dbs[i], dz.sum(0), scale=scale, tol=2e-6)
    # accumulate on top of what is there; bias gradient skipped
    ops.lin_bwd_weight(dys[0], ys[0], xs[0], dws[0], None, ACTS[act], x2=x2s[0], accumulate=True)
    dz = dys[0].double() * _dact64(ys[0].double(), act)
    xc = xs[0].double() if x2s[0] is None else torch.cat([xs[0], x2s[0]], 1).double()
    _close(dws[0], 2 * (dz.t() @ xc), scale=(B ** 0.5) * 6, tol=2e-6)


@pytest.mark.parametrize("B,K1,K2,N", [(128, 256, 0, 1), (128, 3, 1, 256), (200, 64, 0, 48)])
def test_lin_bwd_input_summed_over_items(B, K1, K2, N):
    """Layers fed by one input: one launch returns the sum of their input gradients (items may differ in activation)."""
    from gymrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B + N)
    d, K = "cuda", K1 + K2
    acts = ["none", "clamp", "relu"]
    dys = [torch.randn(B, N, generator=g).to(d) for _ in acts]
    ys = [_act64(torch.randn(B, N, generator=g).double(), a).float().to(d) for a in acts]
    ws = [(torch.randn(N, K, generator=g) / N ** 0.5).to(d) for _ in acts]
    dx, dx2 = ops.lin_bwd_input(dys, ys, ws, [ACTS[a] for a in acts], K1=K1, lo=LO, hi=HI, sum_items=True)
    ref = sum((dy.double() * _dact64(y.double(), a)) @ w.double() for dy, y, w, a in zip(dys, ys, ws, acts))
    _close(dx, ref[:, :K1], scale=float(ref.abs().max()), tol=2e-6)
    if K2:
        _close(dx2, ref[:, K1:], scale=float(ref.abs().max()), tol=2e-6)


def test_fused_linear_module_matches_autograd():
    """gymrl_amd.nn.fused_linears (forward + backward through autograd, gradients written through an armed GradSink)
    against the same layers on torch's own Linear / activation ops in float64."""
    from gymrl_amd import nn as gnn
    from gymrl_amd.flat import GradSink, flatten_module
    torch.manual_seed(3)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a1 = gnn.SmallLinear(5, 64, act="relu")
            self.b1 = gnn.SmallLinear(5, 64, act="relu")
            self.a2 = gnn.SmallLinear(64, 3)
            self.b2 = gnn.SmallLinear(64, 3, act="clamp", clamp=(-0.2, 0.3))

        def forward(self, s, u):
            h, k = gnn.fused_linears([self.a1, self.b1], [s, s], [u, u])
            return gnn.fused_linears([self.a2, self.b2], [h, k])

    net = Net()
    ref = Net().double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    flat, grads = flatten_module(net, torch.device("cuda"))
    sink = GradSink(net)
    s = torch.randn(96, 3, device="cuda")
    u = torch.randn(96, 2, device="cuda", requires_grad=True)
    gnn.FUSED_LINEAR = True
    ya, yb = net(s, u)
    grads.fill_(123.0)                      # whatever the buffer held is overwritten, not accumulated
    sink.arm()
    torch.autograd.backward([ya, yb], [torch.ones_like(ya), 2 * torch.ones_like(yb)])
    sink.collect()
    s64, u64 = s.double().cpu(), u.detach().double().cpu().requires_grad_(True)
    gnn.FUSED_LINEAR = False
    ra, rb = ref(s64, u64)
    gnn.FUSED_LINEAR = True
    torch.autograd.backward([ra, rb], [torch.ones_like(ra), 2 * torch.ones_like(rb)])
    _close(ya.detach(), ra.detach(), tol=2e-6)
    _close(yb.detach(), rb.detach(), tol=2e-6)
    _close(u.grad.detach(), u64.grad, tol=5e-6)
    for (name, p), q in zip(net.named_parameters(), ref.parameters()):
        _close(p.grad, q.grad, scale=float(q.grad.abs().max()) + 1e-3, tol=5e-6)
    # a second backward without arm(): autograd's usual accumulation into .grad (the views), through fresh tensors
    before = [p.grad.clone() for p in net.parameters()]
    ya, yb = net(s, u)
    torch.autograd.backward([ya, yb], [torch.ones_like(ya), 2 * torch.ones_like(yb)])
    for p, b in zip(net.parameters(), before):
        _close(p.grad, 2 * b.double(), scale=float(b.abs().max()) + 1e-3, tol=5e-6)


def test_noisy_dueling_head_matches_autograd():
    """Rainbow's head (two NoisyLinear streams + dueling combination) on the fused launches against the op-by-op
    module path in float64: q, the input gradient and all eight parameter gradients."""
    from gymrl_amd import nn as gnn
    from gymrl_amd.flat import GradSink, flatten_module
    from gymrl_amd.rainbow_dqn_cartpole import DuelingNoisyNetwork
    torch.manual_seed(11)
    net = DuelingNoisyNetwork(4, 3, hidden_dim=64, seed=5)
    ref = DuelingNoisyNetwork(4, 3, hidden_dim=64, seed=5).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    flatten_module(net, torch.device("cuda"))
    sink = GradSink(net)
    x = torch.randn(200, 4, device="cuda")
    for training in (True, False):
        net.train(training)
        ref.train(training)
        q = net(x)
        sink.arm()
        q.backward(torch.arange(600, device="cuda", dtype=torch.float32).view(200, 3) / 600)
        sink.collect()
        # the reference path with the very noise the fused forward used
        gnn.FUSED_LINEAR = False
        try:
            for name in ("advantage", "value"):
                m, r = getattr(net, name), getattr(ref, name)
                r.weight_epsilon.copy_(m.weight_epsilon.double().cpu())
                r.bias_epsilon.copy_(m.bias_epsilon.double().cpu())
                r.reset_noise = lambda: None
            ref.zero_grad()
            qr = ref(x.double().cpu())
            qr.backward(torch.arange(600, dtype=torch.float64).view(200, 3) / 600)
        finally:
            gnn.FUSED_LINEAR = True
        _close(q.detach(), qr.detach(), tol=2e-6)
        for (name, p), r in zip(net.named_parameters(), ref.parameters()):
            want = torch.zeros_like(r) if r.grad is None else r.grad
            _close(p.grad, want, scale=float(want.abs().max()) + 1e-2, tol=5e-6)


def test_wide_linear_at_large_batch_matches_autograd():
    """SmallLinear(128, 256) at 20000 rows (beyond the layer kernels' forward range): library GEMMs forward / input gradient,
    gymrl_lin_bwd_weight for the weight + bias gradient (nn._WideLinear) against float64 autograd."""
    from gymrl_amd.nn import SmallLinear
    torch.manual_seed(3)
    layer = SmallLinear(128, 256)
    ref = torch.nn.Linear(128, 256).double()
    ref.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    x, g = torch.randn(20000, 128), torch.randn(20000, 256)
    x64 = x.double().requires_grad_(True)
    ref(x64).backward(g.double())
    layer = layer.cuda()
    xd = x.cuda().requires_grad_(True)
    y = layer(xd)
    assert type(y.grad_fn).__name__ == "_WideLinearBackward"
    y.backward(g.cuda())
    _close(y.detach(), ref(x64).detach(), tol=5e-6)
    _close(xd.grad, x64.grad, tol=5e-6)
    _close(layer.weight.grad, ref.weight.grad, tol=5e-6)
    _close(layer.bias.grad, ref.bias.grad, tol=5e-6)


def test_lin_bwd_weight_is_deterministic():
    from gymrl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N, K = 3000, 256, 256
    dy, y, x = (torch.randn(B, n, generator=g).cuda() for n in (N, N, K))
    outs = []
    for _ in range(3):
        dw, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
        ops.lin_bwd_weight(dy, y, x, dw, db, ACTS["relu"])
        outs.append((dw.clone(), db.clone()))
    for dw, db in outs[1:]:
        assert torch.equal(dw, outs[0][0]) and torch.equal(db, outs[0][1])


def test_lin_rejects_bad_arguments():
    from gymrl_amd import ops
    x = torch.zeros(4, 8, device="cuda")
    w = torch.zeros(16, 8, device="cuda")
    with pytest.raises(RuntimeError):
        ops.lin_fwd(x, w, None, act=9)
    with pytest.raises((RuntimeError, ValueError)):
        ops.lin_fwd([x] * 5, [w] * 5, [None] * 5)
    with pytest.raises(RuntimeError):
        ops.lin_fwd(x.cpu(), w, None)
