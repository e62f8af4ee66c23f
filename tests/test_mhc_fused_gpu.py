"""PPO-full's rollout forward on the inference kernels (csrc/mhc.hip + csrc/lin.hip) against the torch modules in float64."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _close(got, ref, tol):
    ref = ref.to(got.device)
    err = float((got.detach().double() - ref).abs().max())
    assert err <= tol * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("rate,dim,B", [(2, 128, 4096), (2, 128, 1), (2, 128, 17), (2, 256, 1003), (2, 64, 37), (4, 32, 300)])
def test_mhc_gates_combine_rmsnorm_match_modules(rate, dim, B):
    from gymrl_amd import ops
    from gymrl_amd.ppo_full_lunarlander import ManifoldHyperConnectionFuse, RMSNorm
    torch.manual_seed(rate * 100 + dim)
    fuse = ManifoldHyperConnectionFuse(dim, rate, 10)
    with torch.no_grad():                                   # the module starts with w = 0: make the read-out matter
        fuse.w.normal_(0, 0.5)
        fuse.alpha.copy_(torch.tensor([0.7, -0.4, 0.9]))
        fuse.norm.weight.uniform_(0.5, 1.5)
    h = torch.randn(B, rate, dim) * 2
    ref = ManifoldHyperConnectionFuse(dim, rate, 10).double()
    ref.load_state_dict({k: v.double() for k, v in fuse.state_dict().items()})
    pre64, post64, mix64 = ref.gates(h.double())
    fuse, hd = fuse.cuda(), h.cuda()
    pre, post, mix, read = ops.mhc_gates(hd, fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta, 10)
    _close(pre, pre64, 1e-5)
    _close(post, post64, 1e-5)
    _close(mix, mix64, 1e-5)
    _close(read, torch.bmm(pre64.unsqueeze(1), h.double()).squeeze(1), 1e-5)
    out = torch.randn(B, dim)
    got = ops.mhc_combine(post, mix, out.cuda(), hd)
    want = torch.bmm(post64.unsqueeze(2), out.double().unsqueeze(1)) + torch.bmm(mix64, h.double())
    _close(got, want, 1e-5)
    norm = RMSNorm(dim)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
    _close(ops.rmsnorm(hd, norm.weight.cuda(), norm.eps, n_sum=rate), norm.double()(h.double().sum(1)), 1e-5)
    _close(ops.rmsnorm(out.cuda(), norm.float().weight.cuda(), norm.eps), norm.double()(out.double()), 1e-5)
    if rate == 2 and rate * dim in (256, 512):              # the read-out sums the backward takes
        stats = ops.mhc_gates(hd, fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta, 10, stats=True)[4]
        flat = h.double().reshape(B, -1)
        _close(stats[:, :8], (ref.norm.weight * flat) @ ref.w, 1e-5)
        _close(stats[:, 8], flat.pow(2).sum(1), 1e-5)


@pytest.mark.parametrize("rate,dim,layers", [(2, 128, 2), (4, 32, 1)])
def test_fused_forward_matches_actor_critic(rate, dim, layers):
    from gymrl_amd.ppo_full_lunarlander import ActorCritic, Config
    cfg = Config()
    cfg.mhc_rate, cfg.mhc_dim, cfg.mhc_layers = rate, dim, layers
    torch.manual_seed(5)
    net = ActorCritic(8, 4, config=cfg)
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "w") and hasattr(m, "alpha"):
                m.w.normal_(0, 0.3)
    ref = ActorCritic(8, 4, config=cfg).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    x = torch.randn(1000, 8)
    logits64, value64 = ref(x.double())
    net = net.cuda()
    logits, value = net.forward_fused(x.cuda())
    _close(logits, logits64, 1e-5)
    _close(value, value64, 1e-5)
    lm, vm = net(x.cuda())                                   # and against the float32 modules themselves
    _close(logits, lm.double(), 2e-5)
    _close(value, vm.double(), 2e-5)


@pytest.mark.parametrize("rate,dim,B", [(2, 128, 1000), (4, 32, 257)])
def test_read_and_combine_backward_match_autograd(rate, dim, B):
    """The training pass's fused read / combine launches: forward and every gradient against the torch expressions in
    float64 (bmm form, ppo_full_lunarlander.py:160-165)."""
    from gymrl_amd.ppo_full_lunarlander import _MhcCombine, _MhcRead
    torch.manual_seed(rate + dim)
    pre, post = torch.rand(B, rate), torch.rand(B, rate) * 2
    mix, out, h = torch.rand(B, rate, rate), torch.randn(B, dim), torch.randn(B, rate, dim)
    g_read, g_h = torch.randn(B, dim), torch.randn(B, rate, dim)
    ref = [t.double().requires_grad_(True) for t in (pre, post, mix, out, h)]
    r_read = torch.bmm(ref[0].unsqueeze(1), ref[4]).squeeze(1)
    r_comb = torch.bmm(ref[1].unsqueeze(2), ref[3].unsqueeze(1)) + torch.bmm(ref[2], ref[4])
    torch.autograd.backward([r_read, r_comb], [g_read.double(), g_h.double()])
    dev = [t.cuda().requires_grad_(True) for t in (pre, post, mix, out, h)]
    read = _MhcRead.apply(dev[0], dev[4])
    comb = _MhcCombine.apply(dev[1], dev[2], dev[3], dev[4])
    torch.autograd.backward([read, comb], [g_read.cuda(), g_h.cuda()])
    _close(read, r_read.detach(), 1e-5)
    _close(comb, r_comb.detach(), 1e-5)
    for d, r in zip(dev, ref):
        _close(d.grad, r.grad, 1e-5)


@pytest.mark.parametrize("dim,B", [(128, 3000), (256, 517)])
def test_gates_backward_matches_autograd(dim, B):
    """The training pass's fused gates (forward gymrl_mhc_gates, backward gymrl_mhc_gates_bwd) against the module's torch
    expression under float64 autograd: the three outputs and the gradients of h, norm.weight, w, alpha, beta; and the
    parameter gradients do not depend on the launch (no atomics)."""
    from gymrl_amd.ppo_full_lunarlander import ManifoldHyperConnectionFuse, _MhcGates
    torch.manual_seed(dim)
    fuse = ManifoldHyperConnectionFuse(dim, 2, 10)
    with torch.no_grad():
        fuse.w.normal_(0, 0.3)
        fuse.alpha.copy_(torch.tensor([0.7, -0.4, 0.9]))
        fuse.norm.weight.uniform_(0.5, 1.5)
    h = torch.randn(B, 2, dim)
    gs = [torch.randn(B, 2), torch.randn(B, 2), torch.randn(B, 2, 2)]
    ref = ManifoldHyperConnectionFuse(dim, 2, 10).double()
    ref.load_state_dict({k: v.double() for k, v in fuse.state_dict().items()})
    h64 = h.double().requires_grad_(True)
    torch.autograd.backward(list(ref.gates(h64)), [g.double() for g in gs])
    fuse = fuse.cuda()
    outs = []
    for _ in range(2):
        for p in fuse.parameters():
            p.grad = None
        hd = h.cuda().requires_grad_(True)
        res = _MhcGates.apply(hd, fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta, 10)
        torch.autograd.backward(list(res), [g.cuda() for g in gs])
        outs.append([hd.grad.clone()] + [p.grad.clone() for p in (fuse.norm.weight, fuse.w, fuse.alpha, fuse.beta)])
    for got, want in zip(res, ref.gates(h64.detach())):
        _close(got, want, 1e-5)
    wants = [h64.grad, ref.norm.weight.grad, ref.w.grad, ref.alpha.grad, ref.beta.grad]
    for got, want in zip(outs[0], wants):
        _close(got, want, 2e-5)
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("one_launch", [True, False])
@pytest.mark.parametrize("dim,B", [(128, 2500), (256, 300), (128, 1), (128, 16500), (128, 17)])   # >= 16384 rows: library GEMMs inside
def test_sub_block_node_matches_autograd(dim, B, one_launch, monkeypatch):
    """A whole MHCBlock (two sub-blocks, each ONE autograd node: _MhcSub) against the module's torch expression in float64:
    output, the gradient of h and of every parameter — incl. the read's and the combine's paths into h that
    gymrl_mhc_gates_bwd folds in, and the Linear's gradients written by the node itself."""
    import gymrl_amd.ppo_full_lunarlander as pf
    monkeypatch.setattr(pf, "FUSED_SUB_FORWARD", one_launch)    # gymrl_mhc_sub_forward (D = 128) / the three forward launches
    monkeypatch.setattr(pf, "FUSED_SUB_BACKWARD", one_launch)   # gymrl_mhc_sub_backward (D = 128) / the five backward launches
    torch.manual_seed(dim + B)
    block = pf.MHCBlock(dim, 2, 10)
    with torch.no_grad():
        for m in (block.mhc1, block.mhc2):
            m.w.normal_(0, 0.3)
            m.alpha.copy_(torch.tensor([0.7, -0.4, 0.9]))
            m.norm.weight.uniform_(0.5, 1.5)
    ref = pf.MHCBlock(dim, 2, 10).double()
    ref.load_state_dict({k: v.double() for k, v in block.state_dict().items()})
    h, g = torch.randn(B, 2, dim), torch.randn(B, 2, dim)
    h64 = h.double().requires_grad_(True)
    out64 = ref(h64)
    out64.backward(g.double())
    block = block.cuda()
    hd = h.cuda().requires_grad_(True)
    assert pf.FUSED_SUB
    out = block(hd)
    assert type(out.grad_fn).__name__ == "_MhcSubBackward"
    out.backward(g.cuda())
    _close(out, out64.detach(), 1e-5)
    _close(hd.grad, h64.grad, 3e-5)
    for (k, p), (_, q) in zip(block.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 3e-5)
    first = [p.grad.clone() for p in block.parameters()]     # fixed-order sums: a second pass gives the same bits
    for p in block.parameters():
        p.grad = None
    hd2 = h.cuda().requires_grad_(True)
    block(hd2).backward(g.cuda())
    assert torch.equal(hd2.grad, hd.grad)
    for a, p in zip(first, block.parameters()):
        assert torch.equal(a, p.grad)


@pytest.mark.parametrize("dim,B,silu", [(128, 3000, False), (256, 700, True), (512, 65, True), (96, 130, False), (256, 1, True)])
def test_rmsnorm_node_matches_autograd(dim, B, silu):
    """RMSNorm (optionally of SiLU(x): the MLPs' Linear -> SiLU -> RMSNorm) as one launch each way against float64 autograd."""
    from gymrl_amd.ppo_full_lunarlander import RMSNorm
    torch.manual_seed(dim)
    norm = RMSNorm(dim)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
    ref = RMSNorm(dim).double()
    ref.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    x, g = torch.randn(B, dim) * 2, torch.randn(B, dim)
    x64 = x.double().requires_grad_(True)
    y64 = ref(x64, silu=silu)
    y64.backward(g.double())
    norm = norm.cuda()
    xd = x.cuda().requires_grad_(True)
    y = norm(xd, silu=silu)
    assert type(y.grad_fn).__name__ == "_RmsNormBackward"
    y.backward(g.cuda())
    _close(y, y64.detach(), 1e-5)
    _close(xd.grad, x64.grad, 2e-5)
    _close(norm.weight.grad, ref.weight.grad, 2e-5)


@pytest.mark.parametrize("dim,silu", [(640, False), (640, True), (64, True)])
def test_rmsnorm_forward_other_widths(dim, silu):
    """gymrl_rmsnorm beyond the register-resident widths (D > 512: the two-pass kernel) and below one wave's 64 columns."""
    from gymrl_amd import ops
    torch.manual_seed(dim)
    x, w = torch.randn(33, dim) * 2, torch.rand(dim) + 0.5
    s = torch.nn.functional.silu(x.double()) if silu else x.double()
    want = s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double()
    _close(ops.rmsnorm(x.cuda(), w.cuda(), 1e-6, act=ops.LIN_ACT["silu"] if silu else 0), want, 1e-5)


@pytest.mark.parametrize("B,layers,obs", [(4096, 2, 8), (1, 2, 8), (37, 1, 8), (1000, 4, 16), (50, 0, 3)])
def test_one_launch_policy_forward_matches_actor_critic(B, layers, obs):
    """gymrl_mhc_policy_forward (the whole rollout forward, 16 rows per workgroup) against the float64 modules, and against
    the per-layer inference kernels it replaces."""
    from gymrl_amd.ppo_full_lunarlander import ActorCritic, Config
    cfg = Config()
    cfg.mhc_layers = layers
    torch.manual_seed(11 + B)
    net = ActorCritic(obs, 4, config=cfg)
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "w") and hasattr(m, "alpha"):
                m.w.normal_(0, 0.3)
                m.alpha.copy_(torch.tensor([0.7, -0.4, 0.9]))
                m.norm.weight.uniform_(0.5, 1.5)
        net.actor.mlp[3].weight.normal_(0, 0.1)              # (initialised at std 0.001: make the logits matter)
    ref = ActorCritic(obs, 4, config=cfg).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    x = torch.randn(B, obs)
    logits64, value64 = ref(x.double())
    net = net.cuda()
    out = net.forward_policy(x.cuda())
    assert out is not None
    _close(out[0], logits64, 1e-5)
    _close(out[1], value64, 1e-5)
    fused = net.forward_fused(x.cuda())
    _close(out[0], fused[0].double(), 2e-5)
    _close(out[1], fused[1].double(), 2e-5)
    assert net.forward_inference(x.cuda())[0].shape == (B, 4)
    # the packed image of the wide operands (gymrl_mhc_policy_pack) changes where a weight is read from, not its value
    from gymrl_amd import ops
    d = net._policy_desc()
    d.image = None
    plain = ops.mhc_policy(d, x.cuda())
    img = ops.mhc_policy_pack(d)
    assert img.numel() == 2 * layers * (128 * 128 + 256 * 8) + 2 * 256 * 128
    d.image = img.data_ptr()
    packed = ops.mhc_policy(d, x.cuda())
    d.image = None
    assert torch.equal(plain[0], packed[0]) and torch.equal(plain[1], packed[1])


@pytest.mark.parametrize("B", [1, 16, 100, 5000, 40000])
def test_one_launch_sub_block_forward_matches_its_three_launches(B):
    """gymrl_mhc_sub_forward against gymrl_mhc_gates + gymrl_lin_fwd + gymrl_mhc_combine(SiLU): gates, read-out sums and branch
    sum bit for bit (same arithmetic), z and h' to the Linear's summation order."""
    from gymrl_amd import ops
    torch.manual_seed(B)
    d = "cuda"
    h = torch.randn(B, 2, 128, device=d) * 2
    nw, w = torch.rand(256, device=d) + 0.5, torch.randn(256, 8, device=d) * 0.3
    alpha, beta = torch.tensor([0.7, -0.4, 0.9], device=d), torch.randn(8, device=d) * 0.1
    W, b = torch.randn(128, 128, device=d) * 0.1, torch.randn(128, device=d) * 0.1
    pre, post, mix, stats, read, z, h_out = ops.mhc_sub_forward(h, nw, w, alpha, beta, W, b, 10)
    pre2, post2, mix2, read2, stats2 = ops.mhc_gates(h, nw, w, alpha, beta, 10, stats=True)
    for got, want in ((pre, pre2), (post, post2), (mix, mix2), (stats, stats2), (read, read2)):
        assert torch.equal(got, want)
    z64 = read.double() @ W.double().t() + b.double()
    _close(z, z64, 2e-6)
    _close(h_out, ops.mhc_combine(post, mix, z, h, act=ops.LIN_ACT["silu"]).double(), 1e-6)


@pytest.mark.parametrize("B,g_b,h_b,sum_b", [(1000, False, False, False), (1000, True, False, False), (1000, False, True, True),
                                             (33, True, True, True), (40000, False, False, False)])
def test_one_launch_sub_block_backward_matches_its_launches(B, g_b, h_b, sum_b):
    """gymrl_mhc_sub_backward against the launches it replaces (combine_bwd, the Linear's input gradient, read_bwd, gates_bwd)
    on the same saved tensors — with the upstream gradient / the branch stack broadcast over the branches and the branch-summed
    d_h — at the sums' rounding (2e-5 of the largest value), and bit-identical on a second run (fixed-order sums)."""
    from gymrl_amd import ops
    torch.manual_seed(B)
    dev = "cuda"
    D, sk = 128, 10
    h = torch.randn(B, D, device=dev).unsqueeze(1).repeat(1, 2, 1).contiguous() if h_b else torch.randn(B, 2, D, device=dev)
    g = torch.randn(B, D, device=dev).unsqueeze(1).repeat(1, 2, 1).contiguous() if g_b else torch.randn(B, 2, D, device=dev)
    norm_w = torch.empty(256, device=dev).uniform_(0.5, 1.5)
    w = torch.randn(256, 8, device=dev) * 0.3
    alpha, beta = torch.tensor([0.7, -0.4, 0.9], device=dev), torch.randn(8, device=dev) * 0.1
    W, b = torch.randn(D, D, device=dev) / 11.0, torch.randn(D, device=dev) * 0.1
    pre, post, mix, stats, read, z, _ = ops.mhc_sub_forward(h, norm_w, w, alpha, beta, W, b, sk)
    # the launches
    d_post, d_mix, d_z, _ = ops.mhc_combine_bwd(g, post, mix, z, h, act=ops.LIN_ACT["silu"], want_dh=False)
    d_read, _ = ops.lin_bwd_input(d_z, z, W)
    d_pre, _ = ops.mhc_read_bwd(d_read, pre, h, want_dh=False)
    ref = (d_z,) + tuple(ops.mhc_gates_bwd(h, norm_w, w, alpha, pre, post, mix, stats, d_pre, d_post, d_mix, d_read=d_read, g_out=g))
    if sum_b:
        ref = (ref[0], ref[1].sum(1)) + ref[2:]
    run = lambda: ops.mhc_sub_backward(g[:, 0].contiguous() if g_b else g, h[:, 0].contiguous() if h_b else h, z, pre, post, mix,  # noqa: E731
                                       stats, norm_w, w, alpha, W, sum_branches=sum_b)
    got = run()
    for name, x, y in zip(("d_z", "d_h", "d_norm_w", "d_w", "d_alpha", "d_beta"), got, ref):
        assert x.shape == y.shape, name
        err = float((x.double() - y.double()).abs().max()) / max(1.0, float(y.abs().max()))
        assert err < 2e-5, (name, err)
    again = run()
    for x, y in zip(got, again):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,layers", [(3000, 2), (1, 1), (20000, 1)])
def test_backbone_training_pass_matches_autograd(B, layers):
    """MHCBackbone on the one-launch nodes — the first sub-block fed the un-repeated input projection (its backward returns the
    branches' summed gradient), the final norm summing the branches on load and handing the last sub-block one gradient row
    through a stride-0 view — against the module's torch expression in float64: output, d x, every parameter's gradient."""
    import gymrl_amd.ppo_full_lunarlander as pf
    torch.manual_seed(B + layers)
    net = pf.MHCBackbone(8, 128, 2, layers, 10)
    with torch.no_grad():
        for blk in net.layers:
            for m in (blk.mhc1, blk.mhc2):
                m.w.normal_(0, 0.3)
                m.alpha.copy_(torch.tensor([0.7, -0.4, 0.9]))
                m.norm.weight.uniform_(0.5, 1.5)
        net.final_norm.weight.uniform_(0.5, 1.5)
    ref = pf.MHCBackbone(8, 128, 2, layers, 10).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    x, g = torch.randn(B, 8), torch.randn(B, 128)
    x64 = x.double().requires_grad_(True)
    y64 = ref(x64)
    y64.backward(g.double())
    net = net.cuda()
    xd = x.cuda().requires_grad_(True)
    y = net(xd)
    assert type(y.grad_fn).__name__ == "_RmsNormSumBackward"
    y.backward(g.cuda())
    _close(y, y64.detach(), 1e-5)
    _close(xd.grad, x64.grad, 3e-5)
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 3e-5)


@pytest.mark.parametrize("B,D,n_out,bias", [(3000, 256, 4, True), (1, 256, 1, True), (517, 128, 8, False), (40000, 256, 1, True),
                                            (70, 96, 3, True)])
def test_head_tail_node_matches_autograd(B, D, n_out, bias):
    """SiLU -> RMSNorm -> Linear(D -> n_out) as one launch each way (_NormProj) against the modules in float64: output, d x,
    and the gradients of the norm's weight, the projection and its bias; a second pass gives the same bits."""
    import gymrl_amd.ppo_full_lunarlander as pf
    from gymrl_amd.nn import SmallLinear
    torch.manual_seed(B + D + n_out)
    norm, lin = pf.RMSNorm(D), SmallLinear(D, n_out, bias=bias)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        lin.weight.normal_(0, 0.3)
    x, g = torch.randn(B, D) * 2, torch.randn(B, n_out)
    x64 = x.double().requires_grad_(True)
    w64, W64 = norm.weight.detach().double().requires_grad_(True), lin.weight.detach().double().requires_grad_(True)
    b64 = lin.bias.detach().double().requires_grad_(True) if bias else None
    s = torch.nn.functional.silu(x64)
    y64 = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + norm.eps) * w64) @ W64.t()
    if bias:
        y64 = y64 + b64
    y64.backward(g.double())
    dev = "cuda"
    xd = x.to(dev).requires_grad_(True)
    nw, W = norm.weight.detach().to(dev).requires_grad_(True), lin.weight.detach().to(dev).requires_grad_(True)
    b = lin.bias.detach().to(dev).requires_grad_(True) if bias else None
    y = pf._NormProj.apply(xd, nw, norm.eps, W, b)
    y.backward(g.to(dev))
    _close(y, y64.detach(), 1e-5)
    _close(xd.grad, x64.grad, 2e-5)
    _close(nw.grad, w64.grad, 2e-5)
    _close(W.grad, W64.grad, 2e-5)
    if bias:
        _close(b.grad, b64.grad, 2e-5)
    first = [t.grad.clone() for t in (xd, nw, W)]
    for t in (xd, nw, W):
        t.grad = None
    pf._NormProj.apply(xd, nw, norm.eps, W, b).backward(g.to(dev))
    for a_, t in zip(first, (xd, nw, W)):
        assert torch.equal(a_, t.grad)
