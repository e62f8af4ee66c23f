"""The PPO update's path — hand-written f32-MFMA GEMMs + fused HBM passes, the loss inside the heads pass at hidden 256
(gymrl_amd/ppo_net.FusedActorCriticUpdate.step, ppo_lunarlander.py:274-307) — against torch autograd on the same
module in f32 and in f64, at test sizes and at BASELINE config 2's minibatch (262,144 rows), for every hidden width the
path covers (64 / 128: the narrow-reduction GEMM kernels + the three-pass heads); and the one-pass heads kernel
against the three passes it replaces."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

LOSS_CFG = (0.2, 3.0, 0.5, 0.01)       # clip_eps, dual_clip, value_coef, entropy_coef (ppo_lunarlander.py:36-42)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from gymrl_amd import ops
    assert ops.device_ok()
    return torch.device("cuda:0")


def ppo_loss_torch(logits, values, act, lpo, adv, ret, cfg):
    """ppo_lunarlander.py:110-117, :278-300 restated with torch ops (any dtype)."""
    clip_eps, dual_clip, value_coef, entropy_coef = cfg
    logp_all = torch.log_softmax(logits, dim=-1)
    lp = logp_all.gather(1, act.long().unsqueeze(1)).squeeze(1)
    ent = -(logp_all.exp() * logp_all).sum(-1)
    ratio = torch.exp(lp - lpo)
    s1, s2 = ratio * adv, torch.clamp(ratio, 1 - clip_eps, 1 + clip_eps) * adv
    ms = torch.min(s1, s2)
    pol = -torch.mean(torch.where(adv < 0, torch.max(ms, dual_clip * adv), ms))
    return pol + value_coef * torch.mean((values - ret).pow(2)) - entropy_coef * ent.mean()


def _minibatch(B, dev, seed, D=8, A=4):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, D, device=dev, generator=g)
    act = torch.randint(0, A, (B,), device=dev, generator=g, dtype=torch.int32)
    lpo = -1.386 + 0.2 * torch.randn(B, device=dev, generator=g)
    adv, ret = torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g)
    return x, act, lpo, adv, ret


def _net(dev, seed, hidden=256, D=8, A=4):
    from gymrl_amd import ppo_net
    from gymrl_amd.flat import flatten_module
    from gymrl_amd.ppo_lunarlander import ActorCritic
    torch.manual_seed(seed)
    net = ActorCritic(D, A, hidden)
    with torch.no_grad():
        for p in net.parameters():           # non-zero biases, a policy with opinions: every term is exercised
            if p.dim() == 1:
                p.normal_(0, 0.1)
        net.actor[2].weight.mul_(30.0)
    flatten_module(net, dev, order=ppo_net.LAYOUT)
    return net


@pytest.mark.parametrize("B,hidden,D,A", [(300, 256, 8, 4), (16384, 256, 8, 4), (262144, 256, 8, 4),
                                          (40, 64, 4, 2), (1000, 64, 8, 4), (4097, 128, 3, 4), (65536, 128, 8, 4)])
def test_step_matches_autograd_f32_and_f64(dev, B, hidden, D, A):
    from gymrl_amd import ops, ppo_net
    net = _net(dev, B, hidden, D, A)
    x, act, lpo, adv, ret = _minibatch(B, dev, B + 1, D, A)
    # f32 torch autograd on the same module
    net._flat_grads.zero_()
    lg, vl = net(x)
    ppo_loss_torch(lg, vl.view(-1), act, lpo, adv, ret, LOSS_CFG).backward()
    g32 = {k: p.grad.clone() for k, p in net.named_parameters()}
    # f64 reference of the same arithmetic
    import copy
    net64 = copy.deepcopy(net).double()
    for p in net64.parameters():
        p.grad = None
    lg, vl = net64(x.double())
    loss64 = ppo_loss_torch(lg, vl.view(-1), act, lpo.double(), adv.double(), ret.double(), LOSS_CFG)
    loss64.backward()
    g64 = {k: p.grad.clone() for k, p in net64.named_parameters()}
    del net64, lg, vl
    # the product path
    net._flat_grads.zero_()
    fu = ppo_net.FusedActorCriticUpdate(net, B)
    assert fu.one_pass_heads == (hidden == 256)
    parts = torch.zeros(fu.metric_blocks(B), 5, dtype=torch.float64, device=dev)
    fu.step(x, act, lpo, adv, ret, LOSS_CFG, None, parts)
    worst = 0.0
    for k, p in net.named_parameters():
        scale = float(g64[k].abs().max()) + 1e-30
        e_hip = float((p.grad.double() - g64[k]).abs().max()) / scale
        e_t32 = float((g32[k].double() - g64[k]).abs().max()) / scale
        # within f32 round-off of the f64 gradient, and in the same class as torch's own f32 backward
        assert e_hip <= 2e-6 and e_hip <= 4 * e_t32 + 1e-6, (k, e_hip, e_t32)     # measured: <= 5.5e-7
        worst = max(worst, e_hip)
    m = parts.sum(0).cpu().numpy() / B
    assert abs((m[0] + m[1] - LOSS_CFG[3] * m[2]) - float(loss64.detach())) <= 1e-5 * max(1.0, abs(float(loss64.detach())))
    print(f"B={B} hidden={hidden}: worst relative gradient error vs f64 = {worst:.2e}")


@pytest.mark.parametrize("B,D,A", [(1, 8, 4), (300, 8, 4), (16385, 4, 2), (262144, 8, 4)])
def test_one_finalize_launch_equals_five(dev, B, D, A):
    """step() with the five batch reductions' second halves as ONE launch (gymrl_update_finalize: the default) against the
    five launches behind their producers — every gradient of the flat buffer and the metric partials bit for bit."""
    from gymrl_amd import ppo_net
    outs = []
    for defer in (False, True):
        net = _net(dev, 3, 256, D, A)
        x, act, lpo, adv, ret = _minibatch(B, dev, B + 5, D, A)
        fu = ppo_net.FusedActorCriticUpdate(net, B)
        assert fu.defer_finalize
        fu.defer_finalize = defer
        net._flat_grads.fill_(float("nan"))           # every element is written
        parts = torch.zeros(fu.metric_blocks(B), 5, dtype=torch.float64, device=dev)
        for _ in range(2):                            # the second pass finds the workspaces dirty
            fu.step(x, act, lpo, adv, ret, LOSS_CFG, None, parts)
        assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())
        outs.append((torch.cat([p.grad.reshape(-1) for p in net.parameters()]), parts.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B", [1, 7, 8, 1000, 40000])
def test_heads_loss_one_pass_equals_three_passes(dev, B):
    """gymrl_heads_loss_fwd_bwd vs gymrl_heads_fwd_tanh -> gymrl_ppo_loss_fwd_bwd -> gymrl_heads_bwd on the same
    Zac: dZac bit-identical (same per-row arithmetic); the column reductions and the metric sums group their rows
    differently (8-row batches), so they agree to the round-off of an f32 / f64 sum."""
    from gymrl_amd import ops
    g = torch.Generator(device=dev).manual_seed(B)
    C, A = 256, 4
    Zac = torch.randn(B, 2 * C, device=dev, generator=g)
    bac = 0.1 * torch.randn(2 * C, device=dev, generator=g)
    Wa2, ba2 = torch.randn(A, C, device=dev, generator=g) / 8, 0.1 * torch.randn(A, device=dev, generator=g)
    Wc2, bc2 = torch.randn(1, C, device=dev, generator=g) / 16, 0.1 * torch.randn(1, device=dev, generator=g)
    _, act, lpo, adv, ret = _minibatch(B, dev, B + 3)
    mom = torch.tensor([float(B), 0.3 * B, 1.7 * B], dtype=torch.float64, device=dev)
    ws = ops.mlp_train_workspace(C, 8, A, dev)
    # three passes
    Z3 = Zac.clone()
    logits, value = torch.empty(B, A, device=dev), torch.empty(B, 1, device=dev)
    ops.heads_fwd_tanh(Z3, Wa2, ba2, Wc2, bc2, logits, value, bac, False)
    met3 = torch.zeros(5, dtype=torch.float64, device=dev)
    dl, dv = ops.ppo_loss_fwd_bwd(logits, value.view(-1), act, lpo, adv, ret, LOSS_CFG, adv_moments=mom, metrics_sum=met3)
    out3 = [torch.empty(2 * C, device=dev), torch.empty(A, C, device=dev), torch.empty(A, device=dev),
            torch.empty(1, C, device=dev), torch.empty(1, device=dev)]
    ops.heads_bwd(Z3, dl, dv, Wa2, Wc2, Z3, out3[0], out3[1], out3[2], out3[3], out3[4], ws, True, bac)
    # one pass
    Z1 = Zac.clone()
    out1 = [torch.empty_like(o) for o in out3]
    parts = torch.zeros(ops.heads_loss_blocks(B), 5, dtype=torch.float64, device=dev)
    ops.heads_loss_fwd_bwd(Z1, bac, Wa2, ba2, Wc2, bc2, act, lpo, adv, ret, LOSS_CFG, mom, out1[0], out1[1], out1[2],
                           out1[3], out1[4], parts, ws)
    assert torch.equal(Z1, Z3)
    mag = Z3.abs().sum(0)
    for a, b_, name in zip(out1, out3, ("dbac", "dWa2", "dba2", "dWc2", "dbc2")):
        assert float((a - b_).abs().max()) <= 4e-6 * float(mag.max()) + 1e-7, name
    m1, m3 = parts.sum(0).cpu().numpy(), met3.cpu().numpy()
    assert np.all(np.abs(m1 - m3) <= 1e-10 * np.maximum(1.0, np.abs(m3)) * max(1, B))
