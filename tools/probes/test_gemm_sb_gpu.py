"""The opt-in split-bf16 GEMMs (tools/probes/gemm_sb.hip — a PROBE library since round 5, run with `python -m pytest tools/probes -m gpu`: every f32 operand as three bf16 pieces, six bf16 MFMAs per step, f32
accumulation) against float64 — beside the exact f32-MFMA kernels on the SAME inputs: the mode's error against float64 must
not exceed the exact kernel's class of error (a small multiple, never an order of magnitude), on benign inputs and on
adversarial ones: heavy cancellation, 2^+-60 dynamic range inside one reduction, denormal-adjacent magnitudes, exact
powers of two, signed zeros.  Never compared with the oracle's fmaf chain: this mode is f32-accurate, not bit-exact."""
import numpy as np
import pytest

import os
import sys

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]
import sb  # noqa: E402  (this directory: the probe library's wrappers)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from gymrl_amd import ops
    assert ops.device_ok()
    return torch.device("cuda:0")


def _rel(y, ref):
    return float((y.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-300))


def _cases(B, K, dev, gen):
    """name -> (x [B, K], scale note): the reduction runs over K."""
    x = torch.randn(B, K, device=dev, generator=gen)
    out = {"normal": x}
    # cancellation: pairs of nearly opposite terms; the result is the small residual
    c = torch.randn(B, K, device=dev, generator=gen)
    c[:, 1::2] = -c[:, 0::2] * (1 + 1e-4 * torch.randn(B, K // 2, device=dev, generator=gen))
    out["cancellation"] = c
    # dynamic range: magnitudes 2^-60 .. 2^+60 inside one row
    e = torch.randint(-60, 61, (B, K), device=dev, generator=gen).float()
    out["dynamic_range_2^+-60"] = torch.randn(B, K, device=dev, generator=gen) * torch.exp2(e)
    # tiny magnitudes next to the normal range's floor (products underflow towards denormals)
    out["near_denormal"] = torch.randn(B, K, device=dev, generator=gen) * 2.0 ** -100
    # exact values: powers of two, integers, signed zeros
    p = torch.exp2(torch.randint(-8, 9, (B, K), device=dev, generator=gen).float())
    p[:, ::7] = 0.0
    p[:, 3::7] = -0.0
    out["powers_of_two_and_zeros"] = p
    return out


@pytest.mark.parametrize("N", [256, 512])
def test_split_bf16_forward_is_f32_accurate(dev, N):
    from gymrl_amd import ops
    gen = torch.Generator(device=dev).manual_seed(N)
    B, K = 4096 + 37, 256
    W = torch.randn(N, K, device=dev, generator=gen) / 16
    b = torch.randn(N, device=dev, generator=gen)
    for name, x in _cases(B, K, dev, gen).items():
        ref = x.double() @ W.double().t() + b.double()
        y_ex = ops.linear_fwd(x, W, b, torch.empty(B, N, device=dev), act=False)
        y_sb = torch.full((B + 2, N), float("nan"), device=dev)
        sb.linear_fwd_sb(x, W, b, y_sb[:B], act=False)
        assert bool(torch.isnan(y_sb[B:]).all()), name                 # nothing written past the last row
        e_ex, e_sb = _rel(y_ex, ref), _rel(y_sb[:B], ref)
        assert e_sb <= max(2.0 * e_ex, 3e-7), (name, e_sb, e_ex)
    # tanh epilogue = the exact kernel's (hardware exp2 / rcp on a sum that differs by f32 round-off)
    x = torch.randn(B, K, device=dev, generator=gen)
    h_ex = ops.linear_fwd(x, W, b, torch.empty(B, N, device=dev), act=True)
    h_sb = sb.linear_fwd_sb(x, W, b, torch.empty(B, N, device=dev), act=True)
    assert float((h_ex - h_sb).abs().max()) <= 1e-5


@pytest.mark.parametrize("N", [256, 512])
@pytest.mark.parametrize("with_h", [True, False])
def test_split_bf16_input_gradient_is_f32_accurate(dev, N, with_h):
    from gymrl_amd import ops
    gen = torch.Generator(device=dev).manual_seed(7 * N + with_h)
    B, K = 3000 + 11, 256
    W = torch.randn(N, K, device=dev, generator=gen) / 16
    H = torch.tanh(torch.randn(B, K, device=dev, generator=gen)) if with_h else None
    for name, dy in _cases(B, N, dev, gen).items():
        ref = dy.double() @ W.double()
        if with_h:
            ref = ref * (1 - H.double() ** 2)
        dx_ex = ops.linear_bwd_input(dy, W, H, torch.empty(B, K, device=dev))
        dx_sb = torch.full((B + 3, K), float("nan"), device=dev)
        sb.linear_bwd_input_sb(dy, W, H, dx_sb[:B])
        assert bool(torch.isnan(dx_sb[B:]).all()), name
        e_ex, e_sb = _rel(dx_ex, ref), _rel(dx_sb[:B], ref)
        assert e_sb <= max(2.0 * e_ex, 3e-7), (name, e_sb, e_ex)


def test_split_is_exact_on_the_device(dev):
    """hi + mid + lo == x exactly for every finite f32 whose low bits survive (|x| >= 2^-102): checked through a product
    with the identity — the six-term expansion then reduces to the three pieces of x times 1."""
    from gymrl_amd import ops
    gen = torch.Generator(device=dev).manual_seed(1)
    B = 2048
    x = torch.randn(B, 256, device=dev, generator=gen) * torch.exp2(torch.randint(-40, 41, (B, 256), device=dev, generator=gen).float())
    eye = torch.eye(256, device=dev)
    y = sb.linear_fwd_sb(x, eye, None, torch.empty(B, 256, device=dev), act=False)
    assert torch.equal(y, x)


@pytest.mark.parametrize("N,act", [(256, True), (512, False)])
def test_split_bf16_forward_on_producer_planes_equals_consumer_split(dev, N, act):
    """The producer-planes variant (gymrl_split_planes + gymrl_linear_fwd_sb_planes) runs the consumer-split kernel's products
    in its order on the same bf16 pieces: bit-identical outputs, ragged last tile included, nothing written past it."""
    from gymrl_amd import ops
    gen = torch.Generator(device=dev).manual_seed(N + 1)
    B, K = 8192 + 21, 256
    W = torch.randn(N, K, device=dev, generator=gen) / 16
    b = torch.randn(N, device=dev, generator=gen)
    for name, x in _cases(B, K, dev, gen).items():
        y_sb = sb.linear_fwd_sb(x, W, b, torch.empty(B, N, device=dev), act=act)
        planes = sb.split_planes(x)
        rebuilt = sum((planes[k].view(torch.bfloat16).float() for k in range(3)))
        if name not in ("near_denormal",):                                 # (the third piece of a near-denormal may flush)
            assert torch.equal(rebuilt, x), name                           # the three pieces ARE the float
        y_pl = torch.full((B + 2, N), float("nan"), device=dev)
        sb.linear_fwd_sb_planes(planes, W, b, y_pl[:B], act=act)
        assert bool(torch.isnan(y_pl[B:]).all()), name
        assert torch.equal(y_pl[:B], y_sb), name
