"""GPU parity, off-policy half: replay ring, n-step windows, PER sum-tree, NoisyNet noise,
epsilon-greedy, TD loss, SAC sampling/losses, running normalisation — HIP (through the
C-ABI) vs the CPU oracle and vs the golden fixtures captured from the reference."""
import numpy as np
import pytest

from conftest import load_golden, rel_close

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ring(cap, D, AW, dev):
    return (torch.zeros(cap, D, device=dev), torch.zeros(cap, AW, dtype=torch.int32, device=dev),
            torch.zeros(cap, device=dev), torch.zeros(cap, D, device=dev), torch.zeros(cap, dtype=torch.uint8, device=dev))


def test_replay_append_gather_uniform(dev, oracle):
    from gymrl_amd import ops
    rng = np.random.default_rng(40)
    cap, D, n = 1000, 4, 96
    ring = _ring(cap, D, 1, dev)
    ref = oracle.ReplayRing(cap, D, 1)
    cursor = 0
    for step in range(25):                                   # wraps around twice
        s, s2 = rng.normal(size=(n, D)).astype(np.float32), rng.normal(size=(n, D)).astype(np.float32)
        a = rng.integers(0, 2, size=(n, 1)).astype(np.int32)
        r = rng.normal(size=n).astype(np.float32)
        f = (rng.random(n) < 0.1).astype(np.uint8)
        ops.replay_append(ring, cursor, t(s, dev), t(a, dev), t(r, dev), t(s2, dev), t(f, dev))
        ref.append(s, a, r, s2, f)
        cursor = (cursor + n) % cap
    assert np.array_equal(ring[0].cpu().numpy(), ref.state) and np.array_equal(ring[2].cpu().numpy(), ref.reward)
    idx = ops.uniform_indices(7, 123456789012, ref.size, 512, dev)
    assert np.array_equal(idx.cpu().numpy(), oracle.uniform_indices(7, 123456789012, ref.size, 512))
    assert len(set(idx.cpu().numpy().tolist())) == 512 and 0 <= int(idx.min()) and int(idx.max()) < ref.size   # random.sample: distinct rows
    full = ops.uniform_indices(7, 5, ref.size, ref.size, dev).cpu().numpy()       # B == size: a permutation of the ring
    assert np.array_equal(np.sort(full), np.arange(ref.size)) and np.array_equal(full, oracle.uniform_indices(7, 5, ref.size, ref.size))
    got = ops.replay_gather(ring, idx)
    for g, r in zip(got, ref.gather(idx.cpu().numpy())):
        assert np.array_equal(g.cpu().numpy().astype(np.float64), np.asarray(r).astype(np.float64))


def test_nstep_windows_vs_oracle_and_golden(dev, oracle):
    from gymrl_amd import ops
    rng = np.random.default_rng(41)
    n_steps, N, D, cap, gamma = 5, 37, 4, 400, 0.9
    win = (torch.zeros(n_steps, N, D, device=dev), torch.zeros(n_steps, N, dtype=torch.int32, device=dev),
           torch.zeros(n_steps, N, device=dev), torch.zeros(n_steps, N, D, device=dev),
           torch.zeros(n_steps, N, dtype=torch.uint8, device=dev), torch.zeros(n_steps, N, dtype=torch.uint8, device=dev))
    ring = _ring(cap, D, 1, dev)
    ref_ring, ref_win = oracle.ReplayRing(cap, D), oracle.NStepWindows(n_steps, N, D, gamma)
    cursor = 0
    for step in range(40):
        obs, nxt = rng.normal(size=(N, D)).astype(np.float32), rng.normal(size=(N, D)).astype(np.float32)
        a = rng.integers(0, 2, size=N).astype(np.int32)
        r = rng.normal(size=N).astype(np.float32)
        done = (rng.random(N) < 0.2).astype(np.uint8)
        term = (done & (rng.random(N) < 0.6)).astype(np.uint8)
        emit = ops.nstep_push(win, n_steps, step, gamma, t(obs, dev), t(a, dev), t(r, dev), t(nxt, dev), t(term, dev),
                              t(done, dev), ring, cursor)
        assert emit == ref_win.push(ref_ring, obs, a, r, nxt, term, done)
        if emit:
            cursor = (cursor + N) % cap
    for g, r in zip(ring, (ref_ring.state, ref_ring.action.view(np.int32), ref_ring.reward, ref_ring.next_state, ref_ring.flag)):
        assert np.array_equal(g.cpu().numpy(), r)
    # the reference's single-env stream (golden): N = 1
    g = load_golden("per_nstep")
    cap = int(g["cap"])
    win = (torch.zeros(5, 1, 4, device=dev), torch.zeros(5, 1, dtype=torch.int32, device=dev), torch.zeros(5, 1, device=dev),
           torch.zeros(5, 1, 4, device=dev), torch.zeros(5, 1, dtype=torch.uint8, device=dev), torch.zeros(5, 1, dtype=torch.uint8, device=dev))
    ring = _ring(cap, 4, 1, dev)
    tree = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
    ws = ops.per_workspace(64, dev)
    mx = torch.zeros(1, dtype=torch.float64, device=dev)
    cursor = size = 0
    for s in range(len(g["rew"])):
        emit = ops.nstep_push(win, 5, s, float(g["gamma"]), t(g["obs"][s][None], dev), t(g["act"][s:s + 1], dev),
                              t(g["rew"][s:s + 1], dev), t(g["obs"][s + 1][None], dev), t(g["term"][s:s + 1], dev),
                              t(g["done"][s:s + 1], dev), ring, cursor)
        if emit:
            if size == 0:
                ops.per_update(tree, cap, 1, ws, idx_start=cursor, prio_scalar=1.0)
            else:
                ops.per_max_leaf(tree, cap, mx, ws)
                ops.per_update(tree, cap, 1, ws, idx_start=cursor, prio_scalar_dev=mx)
            cursor, size = (cursor + 1) % cap, min(size + 1, cap)
    assert np.array_equal(ring[2].cpu().numpy(), g["buf_reward"].astype(np.float32))
    assert np.array_equal(ring[3].cpu().numpy().astype(np.float64), g["buf_next"])
    assert np.array_equal(tree.cpu().numpy(), g["tree_after_store"])                 # float64 tree, bit for bit
    idx, prio, w = ops.per_sample(tree, cap, len(g["u"]), size, float(g["beta"]), ws, u=t(g["u"], dev))
    assert np.array_equal(idx.cpu().numpy(), g["index"])                             # integer draws exact
    assert rel_close(w.cpu().numpy(), g["is_weight"], 1e-6) <= 1e-6
    pr = ops.per_priorities(t(g["td"], dev), float(g["alpha"]), 0.01)
    assert np.array_equal(pr.cpu().numpy(), oracle.per_priorities(g["td"], float(g["alpha"]), 0.01))
    ops.per_update(tree, cap, len(g["index2"]), ws, idx=t(g["index2"].astype(np.int32), dev), prio=pr)
    assert rel_close(tree.cpu().numpy(), g["tree_after_update"], 2e-6) <= 2e-6


@pytest.mark.parametrize("cap", [5, 16, 20, 1024, 1000])
def test_sumtree_update_order_exact(dev, oracle, cap):
    """Batched Delta-propagation == the reference's sequential loop, bit for bit, with
    duplicates, non-power-of-two capacity, consecutive (store) ranges that wrap."""
    from gymrl_amd import ops
    rng = np.random.default_rng(cap)
    tree = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
    ref = oracle.SumTree(cap)
    ws = ops.per_workspace(4096, dev)
    for rnd in range(8):
        B = int(rng.integers(1, min(3 * cap, 700)))
        idx = rng.integers(0, cap, size=B).astype(np.int32)
        pr = rng.random(B) * 5 + 1e-3
        ops.per_update(tree, cap, B, ws, idx=t(idx, dev), prio=t(pr, dev))
        ref.update_many(idx=idx, prio=pr)
        assert np.array_equal(tree.cpu().numpy(), ref.tree), rnd
        n = int(rng.integers(1, cap + 1))
        start = int(rng.integers(0, cap))
        ops.per_update(tree, cap, n, ws, idx_start=start, prio_scalar=1.75)
        ref.update_many(idx_start=start, prio_scalar=1.75, B=n)
        assert np.array_equal(tree.cpu().numpy(), ref.tree), rnd
    mx = torch.zeros(1, dtype=torch.float64, device=dev)
    assert ops.per_max_leaf(tree, cap, mx, ws).item() == ref.max_leaf()
    B = 300
    for variant_b in (False, True):
        u = rng.random(B)
        i_ref, p_ref, w_ref = ref.sample(B, cap, 0.55, u=u, variant_b=variant_b)
        idx, prio, w = ops.per_sample(tree, cap, B, cap, 0.55, ws, u=t(u, dev), variant_b=variant_b)
        assert np.array_equal(idx.cpu().numpy(), i_ref) and np.array_equal(prio.cpu().numpy(), p_ref)
        assert rel_close(w.cpu().numpy(), w_ref, 1e-6) <= 1e-6
        i_ref, _, _ = ref.sample(B, cap, 0.55, seed=5, counter=77, variant_b=variant_b)
        idx, _, _ = ops.per_sample(tree, cap, B, cap, 0.55, ws, seed=5, counter=77, variant_b=variant_b)
        assert np.array_equal(idx.cpu().numpy(), i_ref)


@pytest.mark.parametrize("cap,B", [(4096, 513), (5000, 3000), (20000, 8192), (1 << 20, 8192)])
def test_sumtree_large_batches_exact(dev, oracle, cap, B):
    """The LDS-staged and sort-based passes (512 < B <= 8192 with explicit indices; 8192-row consecutive stores)
    keep the reference's per-node addition order: float64 tree bit-identical to the sequential loop, with heavy
    duplication (indices from a narrow range), a non-power-of-two and the Rainbow-sized 2^20 capacity."""
    from gymrl_amd import ops
    rng = np.random.default_rng(cap + B)
    tree = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
    ref = oracle.SumTree(cap)
    ws = ops.per_workspace(8192, dev)
    for rnd, spread in enumerate((cap, max(8, B // 50), cap)):
        start = int(rng.integers(0, cap))
        n = min(8192, cap)
        ops.per_update(tree, cap, n, ws, idx_start=start, prio_scalar=0.5 + rnd)       # N-row store (may wrap)
        ref.update_many(idx_start=start, prio_scalar=0.5 + rnd, B=n)
        assert np.array_equal(tree.cpu().numpy(), ref.tree), ("store", rnd)
        idx = rng.integers(0, spread, size=B).astype(np.int32)
        pr = rng.random(B) * 3 + 1e-3
        ops.per_update(tree, cap, B, ws, idx=t(idx, dev), prio=t(pr, dev))
        ref.update_many(idx=idx, prio=pr)
        assert np.array_equal(tree.cpu().numpy(), ref.tree), ("update", rnd)


def test_per_variant_b_on_the_gpu(dev, oracle):
    """PER variant B (ddqn_per_cartpole.py:67-147) through the C-ABI: max-priority pushes (:113-117), stratified
    sampling that returns TREE indices (:95-108), priorities min(|err| + eps, error_max)^alpha (:142-147) and the
    tree-index update (:75-80) — against the reference's own fixture and, over random rounds with duplicates, the
    oracle's sequential loop (float64 tree bit for bit)."""
    from gymrl_amd import ops
    g = load_golden("per_variant_b")
    cap = int(g["cap"])
    tree = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
    ws = ops.per_workspace(1024, dev)
    mx = torch.zeros(1, dtype=torch.float64, device=dev)
    cursor = 0
    for _ in range(int(g["n_push"])):
        m = float(ops.per_max_leaf(tree, cap, mx, ws).item())
        ops.per_update(tree, cap, 1, ws, idx_start=cursor, prio_scalar=(m if m != 0 else 1.0))
        cursor = (cursor + 1) % cap
    assert np.array_equal(tree.cpu().numpy(), g["tree_after_push"])
    idx, prio, w = ops.per_sample(tree, cap, len(g["u"]), int(g["size"]), float(g["beta"]), ws, u=t(g["u"], dev), variant_b=True)
    assert np.array_equal(idx.cpu().numpy(), g["indices"])                          # tree indices, exact
    assert rel_close(w.cpu().numpy(), g["is_weight"], 1e-6) <= 1e-6
    pr = ops.per_priorities(t(g["errs"].astype(np.float32), dev), float(g["alpha"]), float(g["eps"]), clip=float(g["error_max"]))
    want = oracle.per_priorities(g["errs"], float(g["alpha"]), float(g["eps"]), float(g["error_max"]))
    assert np.array_equal(pr.cpu().numpy(), want)
    ops.per_update(tree, cap, len(g["indices"]), ws, idx=t(g["indices"].astype(np.int32), dev), idx_is_tree=True, prio=pr)
    assert rel_close(tree.cpu().numpy(), g["tree_after_update"], 2e-6) <= 2e-6      # float32 pow: numpy's powf vs exp(a log x)
    # random rounds: clipped priorities + tree-index updates with duplicate leaves vs the oracle's sequential loop
    rng = np.random.default_rng(9)
    for cap2, B in ((20, 64), (1000, 700), (4096, 3000)):
        tree = torch.zeros(2 * cap2 - 1, dtype=torch.float64, device=dev)
        ref = oracle.SumTree(cap2)
        ops.per_update(tree, cap2, cap2, ws if cap2 <= 1024 else ops.per_workspace(8192, dev), idx_start=0, prio_scalar=1.0)
        ref.update_many(idx_start=0, prio_scalar=1.0, B=cap2)
        ws2 = ops.per_workspace(8192, dev)
        for _ in range(4):
            u = rng.random(B)
            i_ref, _, _ = ref.sample(B, cap2, 0.4, u=u, variant_b=True)
            idx, _, _ = ops.per_sample(tree, cap2, B, cap2, 0.4, ws2, u=t(u, dev), variant_b=True)
            assert np.array_equal(idx.cpu().numpy(), i_ref) and i_ref.min() >= cap2 - 1
            err = (rng.normal(size=B) * 2).astype(np.float32)
            pr = ops.per_priorities(t(err, dev), 0.6, 1e-4, clip=1.0)
            p_ref = oracle.per_priorities(err, 0.6, 1e-4, 1.0)
            assert np.array_equal(pr.cpu().numpy(), p_ref) and p_ref.max() <= 1.0
            ops.per_update(tree, cap2, B, ws2, idx=idx, idx_is_tree=True, prio=pr)
            ref.update_many(idx=i_ref, prio=p_ref, idx_is_tree=True)
            assert np.array_equal(tree.cpu().numpy(), ref.tree)


def test_sumtree_golden(dev):
    from gymrl_amd import ops
    g = load_golden("sumtree")
    for k in range(int(g["n_cases"])):
        cap = int(g[f"c{k}_cap"])
        tree = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
        ws = ops.per_workspace(256, dev)
        for idx, pr, snap in zip(g[f"c{k}_ops_idx"], g[f"c{k}_ops_p"], g[f"c{k}_snaps"]):
            m = idx >= 0
            ops.per_update(tree, cap, int(m.sum()), ws, idx=t(idx[m], dev), prio=t(pr[m], dev))
            assert np.array_equal(tree.cpu().numpy(), snap)
        mx = torch.zeros(1, dtype=torch.float64, device=dev)
        assert ops.per_max_leaf(tree, cap, mx, ws).item() == float(g[f"c{k}_max"])


def test_noisy_and_epsilon_greedy(dev, oracle):
    from gymrl_amd import ops
    g = load_golden("noisy")
    for k in range(int(g["n_cases"])):
        nin, nout = g[f"c{k}_raw_in"].size, g[f"c{k}_raw_out"].size
        w, b = torch.empty(nout, nin, device=dev), torch.empty(nout, device=dev)
        ops.noisy_noise(nin, nout, w, b, t(g[f"c{k}_raw_in"], dev), t(g[f"c{k}_raw_out"], dev))
        w_ref, b_ref = oracle.noisy_noise(nin, nout, g[f"c{k}_raw_in"], g[f"c{k}_raw_out"])
        assert np.array_equal(w.cpu().numpy(), w_ref) and np.array_equal(b.cpu().numpy(), b_ref)
        assert rel_close(w.cpu().numpy(), g[f"c{k}_w_eps"], 1e-6) <= 1e-6
        ops.noisy_noise(nin, nout, w, b, seed=3, counter=k)                         # in-kernel Box-Muller
        w_ref, b_ref = oracle.noisy_noise(nin, nout, seed=3, counter=k)
        assert np.array_equal(w.cpu().numpy(), w_ref) and np.array_equal(b.cpu().numpy(), b_ref)
    z = torch.empty(1, 1 << 16, device=dev)
    ops.noisy_noise(1 << 16, 1, z, torch.empty(1, device=dev), seed=4, counter=1)
    raw = (z[0] / z[0, 0].sign().clamp(min=1)).cpu().numpy()                        # f(eps_out)*f(eps_in): sign-sqrt transformed
    assert abs(np.mean(np.sign(raw))) < 0.02
    rng = np.random.default_rng(42)
    q = rng.normal(size=(5000, 2)).astype(np.float32)
    u = rng.random((5000, 2)).astype(np.float32)
    for kw in (dict(u=u), dict(seed=8, counter=99, env_id0=64)):
        ref = oracle.epsilon_greedy(q, 0.3, **kw)
        kw_t = {k: (t(v, dev) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        assert np.array_equal(ops.epsilon_greedy(t(q, dev), 0.3, **kw_t).cpu().numpy(), ref)
    frac = (ops.epsilon_greedy(t(q, dev), 0.3, seed=1, counter=2).cpu().numpy() != q.argmax(1)).mean()
    assert 0.1 < frac < 0.2                                                         # eps/2 of the draws differ from argmax


def test_dqn_td_loss(dev, oracle):
    from gymrl_amd import ops
    rng = np.random.default_rng(43)
    for B, A in ((64, 2), (256, 2), (3000, 6)):
        q, qo, qt = (rng.normal(size=(B, A)).astype(np.float32) for _ in range(3))
        act = rng.integers(0, A, size=B).astype(np.int32)
        rew = rng.normal(size=B).astype(np.float32)
        flag = (rng.random(B) < 0.2).astype(np.float32)
        w = rng.random(B).astype(np.float32)
        for kw in (dict(), dict(q_next_online=qo, w=w)):
            td_ref, dq_ref, l_ref = oracle.dqn_td_loss(q, qt, act, rew, flag, 0.99 ** 5, **kw)
            ls = torch.zeros(1, dtype=torch.float64, device=dev)
            kw_t = {k: t(v, dev) for k, v in kw.items()}
            td, dq = ops.dqn_td_loss(t(q, dev), t(qt, dev), t(act, dev), t(rew, dev), t(flag, dev), 0.99 ** 5, loss_sum=ls, **kw_t)
            assert np.array_equal(td.cpu().numpy(), td_ref) and np.array_equal(dq.cpu().numpy(), dq_ref)
            assert rel_close(ls.cpu().numpy(), l_ref, 1e-9) <= 1e-9
            # against torch autograd on the reference's expression
            qq = torch.tensor(q, requires_grad=True)
            sel = torch.tensor(kw.get("q_next_online", qt)).argmax(1, keepdim=True)
            y = torch.tensor(rew) + (0.99 ** 5) * torch.tensor(qt).gather(1, sel).squeeze(1) * (1 - torch.tensor(flag))
            tdt = qq.gather(1, torch.tensor(act).long()[:, None]).squeeze(1) - y
            loss = (tdt.pow(2) * torch.tensor(kw.get("w", np.ones(B, np.float32)))).mean()
            loss.backward()
            assert np.max(np.abs(dq.cpu().numpy() - qq.grad.numpy())) <= 1e-6 * max(1.0, float(qq.grad.abs().max()))
            assert abs(ls.item() / B - loss.item()) <= 1e-5 * max(1.0, abs(loss.item()))


def test_sac_kernels(dev, oracle):
    from gymrl_amd import ops
    g = load_golden("sac")
    bound = float(g["bound"])
    a_ref, lp_ref = oracle.sac_sample_fwd(g["mean"], g["log_std"], g["eps"], bound)
    a, lp = ops.sac_sample_fwd(t(g["mean"], dev), t(g["log_std"], dev), t(g["eps"], dev), bound)
    assert np.array_equal(a.cpu().numpy(), a_ref) and np.array_equal(lp.cpu().numpy(), lp_ref)
    assert rel_close(a.cpu().numpy(), g["action"]) <= 1e-5 and rel_close(lp.cpu().numpy(), g["logp"]) <= 1e-5
    dm_ref, ds_ref = oracle.sac_sample_bwd(g["mean"], g["log_std"], g["eps"], g["g_action"], g["g_logp"], bound)
    dm, ds = ops.sac_sample_bwd(t(g["mean"], dev), t(g["log_std"], dev), t(g["eps"], dev), t(g["g_action"], dev),
                                t(g["g_logp"], dev), bound)
    assert np.array_equal(dm.cpu().numpy(), dm_ref) and np.array_equal(ds.cpu().numpy(), ds_ref)
    assert np.max(np.abs(dm.cpu().numpy() - g["d_mean"])) <= 2e-5 * np.abs(g["d_mean"]).max()
    rng = np.random.default_rng(44)
    B = 777
    rew, q1, q2, q1n, q2n, lpn, logp = (rng.normal(size=B).astype(np.float32) for _ in range(7))
    q2[:50] = q1[:50]                                            # ties in min(q1, q2)
    done = (rng.random(B) < 0.1).astype(np.float32)
    la = np.log(0.2)
    lad = t(np.array([la]), dev)
    y_ref = oracle.sac_target(rew, done, q1n, q2n, lpn, la, 0.99)
    y = ops.sac_target(t(rew, dev), t(done, dev), t(q1n, dev), t(q2n, dev), t(lpn, dev), lad, 0.99)
    assert rel_close(y.cpu().numpy(), y_ref, 1e-6) <= 1e-6       # exp(log_alpha) in f64: libm vs ocml
    sums = torch.zeros(4, dtype=torch.float64, device=dev)
    d1, d2 = ops.sac_critic_loss(t(q1, dev), t(q2, dev), t(y_ref, dev), sums)
    r1, r2, s_ref = oracle.sac_critic_loss(q1, q2, y_ref)
    assert np.array_equal(d1.cpu().numpy(), r1) and np.array_equal(d2.cpu().numpy(), r2)
    dl, e1, e2 = ops.sac_actor_loss(t(logp, dev), t(q1, dev), t(q2, dev), lad, -1.0, sums)
    rl, f1, f2, s2 = oracle.sac_actor_loss(logp, q1, q2, la, -1.0)
    assert rel_close(dl.cpu().numpy(), rl, 1e-6) <= 1e-6 and np.array_equal(e1.cpu().numpy(), f1)
    assert rel_close(sums.cpu().numpy()[:3], (s_ref + s2)[:3], 1e-7) <= 1e-7
    m, v = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.sac_alpha_step(lad, m, v, sums, B, 3e-4, 1, loss_out=loss)
    la1, m1, v1, l1 = oracle.sac_alpha_step(la, 0.0, 0.0, (s_ref + s2), B, 3e-4, step=1)
    assert abs(lad.item() - la1) <= 1e-12 and abs(loss.item() - l1) <= 1e-9


def test_running_norm_and_reward_scaling(dev, oracle):
    from gymrl_amd import ops
    g = load_golden("normalization")
    stats = torch.zeros(2 + 3 * 8, dtype=torch.float64, device=dev)
    ref_stats = oracle.running_norm_stats(8)
    y = ops.running_norm(t(g["x"], dev), stats)
    y_ref = oracle.running_norm(g["x"], ref_stats)
    assert np.array_equal(y.cpu().numpy(), y_ref) and np.array_equal(stats.cpu().numpy(), ref_stats)
    assert rel_close(y.cpu().numpy(), g["y"], 1e-6) <= 1e-6
    st, R = torch.zeros(5, dtype=torch.float64, device=dev), torch.zeros(60, dtype=torch.float64, device=dev)
    rst, rR = oracle.running_norm_stats(1), np.zeros(60, np.float64)
    r = np.tile(g["r"], 1)
    out = ops.reward_scaling(t(r, dev), t(g["done"], dev), 0.99, R, st)
    ref = oracle.reward_scaling(r, g["done"], 0.99, rR, rst)
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(R.cpu().numpy(), rR)


def test_td3_ddpg_kernels_vs_oracle(dev, oracle):
    """gymrl_noisy_action (both modes, explicit draws and the Philox/Box-Muller stream), gymrl_mse_loss and
    gymrl_neg_mean_loss vs the oracle: bit-exact maps, sums to 1e-12 relative."""
    from gymrl_amd import ops
    rng = np.random.default_rng(8)
    mu = (rng.normal(size=(4097, 1)) * 1.5).astype(np.float32)
    eps = rng.normal(size=(4097, 1))
    for mode, clip in ((0, 0.0), (1, 0.5)):
        got = ops.noisy_action(t(mu, dev), 0.2, 2.0, eps=t(eps, dev), mode=mode, noise_clip=clip)
        assert np.array_equal(got.cpu().numpy(), oracle.noisy_action(mu, 0.2, 2.0, eps=eps, mode=mode, noise_clip=clip))
        got = ops.noisy_action(t(mu, dev), 0.2, 2.0, mode=mode, noise_clip=clip, seed=5, counter=9)
        assert np.array_equal(got.cpu().numpy(), oracle.noisy_action(mu, 0.2, 2.0, mode=mode, noise_clip=clip, seed=5, counter=9))
    noise = ops.noisy_action(torch.zeros(200000, 1, device=dev), 1.0, 100.0, mode=0, seed=1, counter=2).cpu().numpy()
    assert abs(noise.mean()) < 0.01 and abs(noise.std() - 1.0) < 0.01          # the stream is N(0, 1)
    q, y = rng.normal(size=5000).astype(np.float32), rng.normal(size=5000).astype(np.float32)
    s = torch.zeros(1, dtype=torch.float64, device=dev)
    dq = ops.mse_loss(t(q, dev), t(y, dev), s)
    dq_ref, s_ref = oracle.mse_loss(q, y)
    assert np.array_equal(dq.cpu().numpy(), dq_ref) and abs(s.item() - s_ref) <= 1e-12 * abs(s_ref)
    s.zero_()
    dq = ops.neg_mean_loss(t(q, dev), s)
    dq_ref, s_ref = oracle.neg_mean_loss(q)
    assert np.array_equal(dq.cpu().numpy(), dq_ref) and abs(s.item() - s_ref) <= 1e-9


def test_dsac_kernels_vs_oracle(dev, oracle):
    """gymrl_dsac_target / _critic_loss / _actor_loss / _alpha_step vs the oracle: element maps bit-exact
    (same det_expf / det_logf), sums to 1e-12 relative, the scalar Adam step bit-exact over several steps."""
    from gymrl_amd import ops
    rng = np.random.default_rng(12)
    for B, A in ((1, 2), (257, 2), (5000, 4), (4096, 8)):
        z = rng.normal(size=(B, A)) * 2
        p = (np.exp(z) / np.exp(z).sum(1, keepdims=True)).astype(np.float32)
        q1, q2 = (rng.normal(size=(B, A)).astype(np.float32) for _ in range(2))
        rew = rng.normal(size=B).astype(np.float32)
        done = (rng.random(B) < 0.2).astype(np.float32)
        act = rng.integers(0, A, B).astype(np.int32)
        la = np.float32(np.log(0.2))
        la_t = t(np.array([la]), dev)
        y = ops.dsac_target(t(p, dev), t(q1, dev), t(q2, dev), t(rew, dev), t(done, dev), la_t, 0.9)
        y_ref = oracle.dsac_target(p, q1, q2, rew, done, float(la), 0.9)
        assert np.array_equal(y.cpu().numpy(), y_ref), (B, A)
        s = torch.zeros(2, dtype=torch.float64, device=dev)
        d1, d2 = ops.dsac_critic_loss(t(q1, dev), t(q2, dev), t(act, dev), y, s)
        r1, r2, rs = oracle.dsac_critic_loss(q1, q2, act, y_ref)
        assert np.array_equal(d1.cpu().numpy(), r1) and np.array_equal(d2.cpu().numpy(), r2)
        assert np.all(np.abs(s.cpu().numpy() - rs) <= 1e-12 * np.abs(rs))
        s.zero_()
        dp = ops.dsac_actor_loss(t(p, dev), t(q1, dev), t(q2, dev), la_t, s)
        rp, rs = oracle.dsac_actor_loss(p, q1, q2, float(la))
        assert np.array_equal(dp.cpu().numpy(), rp)
        assert np.all(np.abs(s.cpu().numpy() - rs) <= 1e-12 * np.maximum(1.0, np.abs(rs)))
        m, v = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        loss = torch.zeros(1, dtype=torch.float64, device=dev)
        cur, mm, vv = float(la), 0.0, 0.0
        for step in range(1, 5):
            ops.dsac_alpha_step(la_t, m, v, s, B, -1.0, 1e-3, step, loss_out=loss)
            cur, mm, vv, lo = oracle.dsac_alpha_step(cur, mm, vv, rs, B, -1.0, 1e-3, step)
            assert la_t.item() == np.float32(cur) and m.item() == np.float32(mm) and v.item() == np.float32(vv)
            assert abs(loss.item() - lo) <= 1e-12 * max(1.0, abs(lo))


@pytest.mark.parametrize("kind_name", ["CARTPOLE", "PENDULUM"])
def test_env_abandon_vs_oracle(dev, oracle, kind_name):
    """gymrl_env_abandon (a trainer's `for step in range(cfg.max_steps)` below the env's TimeLimit): envs whose episode
    reached the cap restart without a done flag; observations, flags, returns and lengths equal the oracle's, and the
    following steps continue bit for bit on the new episodes' Philox streams."""
    from gymrl_amd import ops
    kind = getattr(ops, kind_name)
    okind = getattr(oracle, kind_name)
    n, seed, cap = 96, 21, 7
    D = 4 if kind_name == "CARTPOLE" else 3
    env = oracle.Env(okind, n, seed=seed)
    o_ref = env.reset()
    state = ops.env_state(kind, n, dev)
    obs, rew = torch.empty(n, D, device=dev), torch.empty(n, device=dev)
    term, trunc, done = (torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(3))
    ops.env_reset(kind, state, n, seed, 0, obs)
    assert np.array_equal(obs.cpu().numpy(), o_ref)
    rng = np.random.default_rng(2)
    hits = 0
    for _ in range(40):
        act = rng.integers(0, 2, size=n).astype(np.int32) if kind_name == "CARTPOLE" else rng.uniform(-2, 2, size=(n, 1)).astype(np.float32)
        r = env.step(act)
        ops.env_step(kind, state, n, seed, 0, t(act, dev), obs, rew, term, trunc, done_out=done)
        assert np.array_equal(obs.cpu().numpy(), r["obs"]) and np.array_equal(done.cpu().numpy(), r["done"])
        o2, flag, ep_ret, ep_len = env.abandon(cap, r["obs"])
        g_ret, g_len = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        ops.env_abandon(kind, state, n, seed, 0, cap, obs, done, g_ret, g_len)
        assert np.array_equal(obs.cpu().numpy(), o2)
        assert np.array_equal(done.cpu().numpy(), r["done"] | flag)
        assert np.array_equal(g_len.cpu().numpy(), ep_len) and np.allclose(g_ret.cpu().numpy(), ep_ret, rtol=0, atol=0)
        assert np.all(ep_len[flag.astype(bool)] == cap)
        hits += int(flag.sum())
    assert hits > n


@pytest.mark.gpu
@pytest.mark.parametrize("cap,B", [(1 << 20, 256), (5000, 512), (1, 1), (300, 37)])
def test_per_update_td_equals_its_three_calls(dev, cap, B):
    """gymrl_per_update_td (update_priorities from the TD errors + the next store's priority_max, two launches) leaves the
    tree and the maximum with the bits of gymrl_per_priorities -> gymrl_per_update -> gymrl_per_max_leaf — duplicates in the
    batch included — and its ticket back at zero."""
    from gymrl_amd import ops
    g = torch.Generator().manual_seed(cap + B)
    tree_a = torch.zeros(2 * cap - 1, dtype=torch.float64, device=dev)
    ws = ops.per_workspace(max(8192, cap), dev)
    n0 = min(cap, 4096)
    ops.per_update(tree_a, cap, n0, ws, idx_start=0, prio=torch.rand(n0, generator=g, dtype=torch.float64).to(dev) + 0.1)
    tree_b = tree_a.clone()
    mx_a, mx_b = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    for rep in range(3):
        idx = torch.randint(0, min(cap, n0), (B,), generator=g).to(torch.int32).to(dev)
        if B > 8:
            idx[5] = idx[2]
            idx[B - 1] = idx[2]
        td = (torch.randn(B, generator=g) * 3.0).to(dev)
        ops.per_update(tree_a, cap, B, ws, idx=idx, prio=ops.per_priorities(td, 0.6, 0.01))
        ops.per_max_leaf(tree_a, cap, mx_a, ws)
        ops.per_update_td(tree_b, cap, idx, td, 0.6, 0.01, ws, max_out=mx_b, ticket=ticket)
        assert torch.equal(tree_a, tree_b), rep
        assert torch.equal(mx_a, mx_b), rep
        assert int(ticket.item()) == 0
