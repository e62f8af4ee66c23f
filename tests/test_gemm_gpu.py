"""Hand-written f32-MFMA GEMMs of the PPO update (gymrl_linear_fwd / _bwd_input / _bwd_weight,
ppo_lunarlander.py:67-84, :110-117, :303) through the C-ABI: bit for bit against the oracle's fmaf
chains in the documented accumulation order, and against an fp64 reference at the bench's minibatch size."""
import numpy as np
import pytest

from conftest import rel_close

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from gymrl_amd import ops
    assert ops.device_ok()
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _data(B, N, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(B, 256)).astype(np.float32)
    W = (rng.normal(size=(N, 256)) / 16).astype(np.float32)         # asymmetric: transposes cannot hide
    b = rng.normal(size=N).astype(np.float32)
    return rng, x, W, b


@pytest.mark.parametrize("B,N", [(1, 256), (37, 256), (300, 256), (257, 512), (2049, 256)])
def test_linear_fwd_bit_exact(dev, oracle, B, N):
    from gymrl_amd import ops
    _, x, W, b = _data(B, N, B + N)
    y = torch.full((B, N), float("nan"), device=dev)
    ops.linear_fwd(t(x, dev), t(W, dev), t(b, dev), y, act=False)
    h = torch.empty(B, N, device=dev)
    ops.linear_fwd(t(x, dev), t(W, dev), t(b, dev), h, act=True)
    want = oracle.linear_fwd(x, W, b)
    assert np.array_equal(y.cpu().numpy(), want)
    # tanh epilogue = 1 - 2 / (exp2(x * 2 log2 e) + 1) on the hardware exp2 / rcp units: absolute error
    assert np.max(np.abs(h.cpu().numpy().astype(np.float64) - np.tanh(want.astype(np.float64)))) <= 4e-7


@pytest.mark.parametrize("B,N", [(1, 256), (95, 256), (300, 512), (1030, 256), (1100, 512)])
@pytest.mark.parametrize("with_h", [True, False])
def test_linear_bwd_input_bit_exact(dev, oracle, B, N, with_h):
    from gymrl_amd import ops
    rng, _, W, _ = _data(B, N, 7 * B + N)
    dy = rng.normal(size=(B, N)).astype(np.float32)
    H = np.tanh(rng.normal(size=(B, 256))).astype(np.float32) if with_h else None
    dx = torch.full((B + 3, 256), float("nan"), device=dev)            # 3 guard rows: nothing may be written past B
    ops.linear_bwd_input(t(dy, dev), t(W, dev), None if H is None else t(H, dev), dx[:B])
    assert np.array_equal(dx[:B].cpu().numpy(), oracle.linear_bwd_input(dy, W, H))
    assert bool(torch.isnan(dx[B:]).all())


@pytest.mark.parametrize("B,N", [(1, 256), (2, 512), (133, 256), (1501, 256), (700, 512)])
def test_linear_bwd_weight_bit_exact(dev, oracle, B, N):
    from gymrl_amd import ops
    rng, x, _, _ = _data(B, N, 13 * B + N)
    dy = rng.normal(size=(B, N)).astype(np.float32)
    slices, rps = ops.linear_bwd_weight_geometry(B, N)
    assert slices * rps >= B and (slices - 1) * rps < B and rps % 2 == 0
    dW = torch.full((N, 256), float("nan"), device=dev)
    db = torch.full((N,), float("nan"), device=dev)
    ops.linear_bwd_weight(t(dy, dev), t(x, dev), dW, ops.gemm_workspace(dev), db)
    assert np.array_equal(dW.cpu().numpy(), oracle.linear_bwd_weight(dy, x, slices, rps))
    assert np.array_equal(db.cpu().numpy(), oracle.linear_bwd_bias(dy, slices, rps))


@pytest.mark.parametrize("K,N", [(64, 64), (64, 128), (128, 128), (128, 256)])
@pytest.mark.parametrize("B", [1, 31, 33, 700, 4101])
def test_narrow_linear_fwd_bit_exact(dev, oracle, K, N, B):
    """The 64- / 128-long reductions (gemm_ns_kernel: row tiles staged through LDS; PPO-full's Linear layers at
    262,144-row micro-batches, ActorCritic at hidden 64 / 128): the same accumulation order as the 256-wide kernels,
    so the oracle's fmaf chain restates them bit for bit — ragged row counts, rows past the end untouched."""
    from gymrl_amd import ops
    rng = np.random.default_rng(1000 * K + N + B)
    x = rng.normal(size=(B, K)).astype(np.float32)
    W = (rng.normal(size=(N, K)) / 8).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    y = torch.full((B + 2, N), float("nan"), device=dev)
    ops.linear_fwd(t(x, dev), t(W, dev), t(b, dev), y[:B], act=False)
    want = oracle.linear_fwd(x, W, b)
    assert np.array_equal(y[:B].cpu().numpy(), want) and bool(torch.isnan(y[B:]).all())
    y0 = torch.empty(B, N, device=dev)
    ops.linear_fwd(t(x, dev), t(W, dev), None, y0, act=False)                 # no bias
    assert np.array_equal(y0.cpu().numpy(), oracle.linear_fwd(x, W, None))
    h = torch.empty(B, N, device=dev)
    ops.linear_fwd(t(x, dev), t(W, dev), t(b, dev), h, act=True)
    assert np.max(np.abs(h.cpu().numpy().astype(np.float64) - np.tanh(want.astype(np.float64)))) <= 4e-7


@pytest.mark.parametrize("N,K", [(64, 64), (128, 64), (128, 128), (256, 128)])
@pytest.mark.parametrize("B", [1, 95, 1030])
@pytest.mark.parametrize("with_h", [True, False])
def test_narrow_linear_bwd_input_bit_exact(dev, oracle, N, K, B, with_h):
    from gymrl_amd import ops
    rng = np.random.default_rng(77 * N + K + B)
    W = (rng.normal(size=(N, K)) / 8).astype(np.float32)
    dy = rng.normal(size=(B, N)).astype(np.float32)
    H = np.tanh(rng.normal(size=(B, K))).astype(np.float32) if with_h else None
    dx = torch.full((B + 3, K), float("nan"), device=dev)
    ops.linear_bwd_input(t(dy, dev), t(W, dev), None if H is None else t(H, dev), dx[:B])
    assert np.array_equal(dx[:B].cpu().numpy(), oracle.linear_bwd_input(dy, W, H))
    assert bool(torch.isnan(dx[B:]).all())


def test_narrow_gemms_at_micro_batch_size_vs_fp64(dev):
    """PPO-full's micro-batch (262,144 rows) through the 128-wide kernels, against fp64 GEMMs on the device."""
    from gymrl_amd import ops
    B = 262144
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(B, 128, device=dev, generator=g)
    for N in (128, 256):
        W = torch.randn(N, 128, device=dev, generator=g) / 11
        b = torch.randn(N, device=dev, generator=g)
        y = ops.linear_fwd(x, W, b, torch.empty(B, N, device=dev), act=False)
        ref = x.double() @ W.double().t() + b.double()
        assert float((y.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        dy = torch.randn(B, N, device=dev, generator=g)
        dx = ops.linear_bwd_input(dy, W, None, torch.empty(B, 128, device=dev))
        ref = dy.double() @ W.double()
        assert float((dx.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        del ref


def test_gemms_at_minibatch_size_vs_fp64(dev):
    """B = 262,144 (BASELINE config 2's minibatch): every output against an fp64 GEMM on the device; the error is
    f32 round-off of a 256 / 512 / 262,144-term chain, and not worse than the library's f32 GEMM."""
    from gymrl_amd import ops
    B = 262144
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(B, 256, device=dev, generator=g)
    dy = torch.randn(B, 512, device=dev, generator=g)
    W = torch.randn(512, 256, device=dev, generator=g) / 16
    b = torch.randn(512, device=dev, generator=g)
    ws = ops.gemm_workspace(dev)

    def err(got, want64):
        return float(((got.double() - want64).abs().max() / want64.abs().max()))

    y = torch.empty(B, 512, device=dev)
    ops.linear_fwd(x, W, b, y, act=False)
    want = x.double() @ W.double().t() + b.double()
    e_hip, e_lib = err(y, want), err(torch.addmm(b, x, W.t()), want)
    assert e_hip <= 2e-6 and e_hip <= 2 * e_lib + 1e-7, (e_hip, e_lib)
    del want
    H = torch.tanh(torch.randn(B, 256, device=dev, generator=g))
    dx = torch.empty(B, 256, device=dev)
    ops.linear_bwd_input(dy, W, H, dx)
    want = (dy.double() @ W.double()) * (1 - H.double() ** 2)
    assert err(dx, want) <= 2e-6
    del want
    dW, db = torch.empty(512, 256, device=dev), torch.empty(512, device=dev)
    ops.linear_bwd_weight(dy, x, dW, ws, db)
    want = dy.double().t() @ x.double()
    e_hip, e_lib = err(dW, want), err(dy.t() @ x, want)
    assert e_hip <= 1e-4 and e_hip <= 4 * e_lib + 1e-6, (e_hip, e_lib)     # 262,144-term f32 sums
    assert float((db.double() - dy.double().sum(0)).abs().max()) <= 1e-5 * float(dy.double().abs().sum(0).max())
