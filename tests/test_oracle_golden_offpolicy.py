"""Pin the off-policy half of the CPU oracle (SumTree, n-step PER buffer, NoisyLinear noise,
SAC sampling, running normalisation) to golden vectors captured from the reference's classes."""
import numpy as np

from conftest import load_golden, rel_close


def test_sumtree_traces_exact(oracle):
    g = load_golden("sumtree")
    for k in range(int(g["n_cases"])):
        cap = int(g[f"c{k}_cap"])
        tree = oracle.SumTree(cap)
        for idx, pr, snap in zip(g[f"c{k}_ops_idx"], g[f"c{k}_ops_p"], g[f"c{k}_snaps"]):
            m = idx >= 0
            tree.update_many(idx=idx[m], prio=pr[m])
            assert np.array_equal(tree.tree, snap)                       # float64 array, bit for bit
        L = oracle.lib()
        L.orc_tree_get_index.restype = __import__("ctypes").c_int64
        for v, gi, gp in zip(g[f"c{k}_v"], g[f"c{k}_get_idx"], g[f"c{k}_get_prio"]):
            import ctypes as C
            pr = C.c_double()
            p = L.orc_tree_get_index(tree.tree.ctypes.data_as(C.c_void_p), C.c_int64(cap), C.c_double(float(v)), C.byref(pr))
            assert p - cap + 1 == gi and pr.value == gp
        assert tree.max_leaf() == float(g[f"c{k}_max"])


def test_nstep_per_buffer_matches_reference(oracle):
    g = load_golden("per_nstep")
    cap, n, gamma = int(g["cap"]), int(g["n_steps"]), float(g["gamma"])
    ring = oracle.ReplayRing(cap, 4)
    win = oracle.NStepWindows(n, 1, 4, gamma)
    tree = oracle.SumTree(cap)
    size = 0
    for t in range(len(g["rew"])):
        cursor = ring.cursor
        emitted = win.push(ring, g["obs"][t][None], g["act"][t:t + 1], g["rew"][t:t + 1], g["obs"][t + 1][None],
                           g["term"][t:t + 1], g["done"][t:t + 1])
        if emitted:                                                      # :201-205
            pr = 1.0 if size == 0 else tree.max_leaf()
            tree.update_many(idx_start=cursor, prio_scalar=pr, B=1)
            size = min(size + 1, cap)
    assert size == int(g["size"]) and ring.cursor == int(g["count"])
    assert np.array_equal(ring.state.astype(np.float64), g["buf_state"])
    assert np.array_equal(ring.next_state.astype(np.float64), g["buf_next"])
    assert np.array_equal(ring.action[:, 0].astype(np.float64), g["buf_action"][:, 0])
    assert np.array_equal(ring.flag.astype(np.float64), g["buf_terminal"])
    assert np.array_equal(ring.reward, g["buf_reward"].astype(np.float32))   # float64 R, float32 at use
    assert np.array_equal(tree.tree, g["tree_after_store"])
    idx, prio, w = tree.sample(len(g["u"]), size, float(g["beta"]), u=g["u"])
    assert np.array_equal(idx, g["index"])
    assert rel_close(w, g["is_weight"], 1e-6) <= 1e-6
    st, _, rw, _, _ = ring.gather(idx)
    assert np.array_equal(rw, g["batch_reward"]) and np.array_equal(st, g["batch_state"])
    pr = oracle.per_priorities(g["td"], float(g["alpha"]), 0.01)
    tree.update_many(idx=g["index2"], prio=pr)
    assert rel_close(tree.tree, g["tree_after_update"], 2e-6) <= 2e-6    # float32 pow: numpy's powf vs exp(a*log x)


def test_per_variant_b_matches_reference(oracle):
    g = load_golden("per_variant_b")
    cap = int(g["cap"])
    tree = oracle.SumTree(cap)
    cursor = 0
    for t in range(int(g["n_push"])):                                    # ddqn_per_cartpole.py:113-117
        mx = tree.max_leaf()
        tree.update_many(idx_start=cursor, prio_scalar=(mx if mx != 0 else 1.0), B=1)
        cursor = (cursor + 1) % cap
    assert np.array_equal(tree.tree, g["tree_after_push"])
    idx, prio, w = tree.sample(len(g["u"]), int(g["size"]), float(g["beta"]), u=g["u"], variant_b=True)
    assert np.array_equal(idx, g["indices"])
    assert rel_close(w, g["is_weight"], 1e-6) <= 1e-6
    pr = oracle.per_priorities(g["errs"], float(g["alpha"]), float(g["eps"]), float(g["error_max"]))
    tree.update_many(idx=g["indices"], prio=pr, idx_is_tree=True)
    assert rel_close(tree.tree, g["tree_after_update"], 2e-6) <= 2e-6


def test_noisy_noise_matches_reference(oracle):
    g = load_golden("noisy")
    for k in range(int(g["n_cases"])):
        nin, nout = g[f"c{k}_raw_in"].size, g[f"c{k}_raw_out"].size
        w, b = oracle.noisy_noise(nin, nout, g[f"c{k}_raw_in"], g[f"c{k}_raw_out"])
        assert rel_close(b, g[f"c{k}_b_eps"], 1e-6) <= 1e-6            # torch's vectorised sqrt differs in the last ulp
        assert rel_close(w, g[f"c{k}_w_eps"], 1e-6) <= 1e-6


def test_sac_sample_matches_autograd(oracle):
    g = load_golden("sac")
    act, logp = oracle.sac_sample_fwd(g["mean"], g["log_std"], g["eps"], float(g["bound"]))
    assert rel_close(act, g["action"]) <= 1e-5 and rel_close(logp, g["logp"]) <= 1e-5
    dm, ds = oracle.sac_sample_bwd(g["mean"], g["log_std"], g["eps"], g["g_action"], g["g_logp"], float(g["bound"]))
    assert np.max(np.abs(dm - g["d_mean"])) <= 2e-5 * np.abs(g["d_mean"]).max()
    assert np.max(np.abs(ds - g["d_log_std"])) <= 2e-5 * np.abs(g["d_log_std"]).max()


def test_normalization_matches_reference(oracle):
    g = load_golden("normalization")
    stats = oracle.running_norm_stats(8)
    y = oracle.running_norm(g["x"], stats)
    assert rel_close(y, g["y"], 1e-6) <= 1e-6
    assert stats[0] == int(g["n"])
    assert np.array_equal(stats[2:10].astype(np.float32), g["mean"].astype(np.float32))
    assert rel_close(stats[10:18], g["S"], 1e-12) <= 1e-12 and rel_close(stats[18:26], g["std"], 1e-12) <= 1e-12
    y_eval = oracle.running_norm(g["x"][3:4], stats.copy(), update=False)
    assert rel_close(y_eval[0], g["y_eval"], 1e-6) <= 1e-6
    # RewardScaling: one stream, reset on done; first output of a fresh scaler is r/(r+1e-8) ~ +1
    st, R = oracle.running_norm_stats(1), np.zeros(1, np.float64)
    outs = []
    for r, d in zip(g["r"], g["done"]):
        outs.append(oracle.reward_scaling(np.array([r]), np.array([d]), float(g["gamma"]), R, st)[0])
    assert rel_close(np.array(outs), g["r_scaled"], 1e-6) <= 1e-6
    assert abs(outs[0] - 1.0) < 1e-6


def test_td3_ddpg_noise_and_losses(oracle):
    """SURVEY 8f.3 kernels' restatement vs the reference: exploration noise of select_action (numpy float64
    arithmetic, stored float32), and the TD target / MSE / -mean pieces on the golden batch are exercised by
    the GPU trainer test; here the elementary maps."""
    g = load_golden("td3_ddpg")
    for name, std in (("ddpg", 0.1), ("td3", 0.1)):
        det = g[f"{name}_sel_det"].astype(np.float32)
        got = oracle.noisy_action(det, std * 2.0, 2.0, eps=g[f"{name}_sel_eps"], mode=0)
        assert np.array_equal(got, g[f"{name}_sel_action"].astype(np.float32)), name
    # target-policy smoothing (td3_pendulum.py:191-196) against torch's float32 clamp arithmetic
    import torch
    mu = torch.linspace(-2, 2, 41)
    eps = torch.linspace(-4, 4, 41).flip(0).double()
    want = (mu + (eps.float() * 0.2).clamp(-0.5, 0.5)).clamp(-2.0, 2.0).numpy()
    assert np.array_equal(oracle.noisy_action(mu.numpy(), 0.2, 2.0, eps=eps.numpy(), mode=1, noise_clip=0.5), want)
    q, y = torch.randn(50), torch.randn(50)
    q.requires_grad_(True)
    loss = torch.nn.functional.mse_loss(q, y)
    loss.backward()
    dq, s = oracle.mse_loss(q.detach().numpy(), y.numpy())
    assert np.allclose(dq, q.grad.numpy(), rtol=1e-6, atol=1e-8) and abs(s / 50 - loss.item()) <= 1e-6
    dq, s = oracle.neg_mean_loss(q.detach().numpy())
    assert np.array_equal(dq, np.full(50, -1.0 / 50, np.float32)) and abs(-s / 50 + q.mean().item()) <= 1e-6


def test_dsac_pieces_match_reference(oracle):
    """SURVEY 8f.3 discrete SAC: soft-Bellman target, both critic losses + dL/dQ, actor loss + dL/dprobs and the
    float32 log_alpha Adam step vs tensors computed by the reference's own networks / optimiser (dsac.npz)."""
    g = load_golden("dsac")
    la0 = float(g["log_alpha0"][0])
    y = oracle.dsac_target(g["k_next_probs"], g["k_next_q1"], g["k_next_q2"], g["rewards"], g["dones"].astype(np.float32),
                           la0, float(g["gamma"]))
    assert rel_close(y, g["k_target_q"], 1e-6) <= 1e-6
    d1, d2, s = oracle.dsac_critic_loss(g["k_q1"], g["k_q2"], g["actions"], g["k_target_q"])
    B = len(y)
    assert np.allclose(d1, g["k_dq1"], rtol=1e-6, atol=1e-9) and np.allclose(d2, g["k_dq2"], rtol=1e-6, atol=1e-9)
    assert rel_close(s / B, g["k_critic_losses"], 1e-6) <= 1e-6
    dp, sa = oracle.dsac_actor_loss(g["k_probs"], g["k_q1"], g["k_q2"], la0)
    assert np.allclose(dp, g["k_dprobs"], rtol=2e-6, atol=1e-8)
    assert abs(sa[0] / B - float(g["k_actor_loss"])) <= 1e-6 and abs(sa[1] / B - float(g["k_entropy_mean"])) <= 1e-6
    # first alpha step of the reference run: the actor has moved by then, so only the Adam arithmetic is pinned:
    # with m = v = 0, step 1 moves log_alpha by -lr * sign(g) (up to eps)
    la1, m, v, loss = oracle.dsac_alpha_step(la0, 0.0, 0.0, sa, B, float(g["target_entropy"]), 1e-3, 1)
    assert abs(loss - float(g["losses"][0][3])) <= 2e-3          # same entropy up to one actor step
    assert abs(la1 - float(g["log_alphas"][0])) <= 1e-6
    import torch
    p = torch.tensor([la0], requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-3)
    mm = vv = 0.0
    cur = la0
    for step, ent in enumerate((0.61, 0.35, 0.5), 1):
        opt.zero_grad()
        (p.exp() * (torch.tensor(ent) - float(g["target_entropy"]))).sum().backward()
        opt.step()
        cur, mm, vv, _ = oracle.dsac_alpha_step(cur, mm, vv, np.array([0.0, ent * 8]), 8, float(g["target_entropy"]), 1e-3, step)
        assert abs(cur - p.item()) <= 2e-7, step


def test_uniform_indices_are_a_sample_without_replacement(oracle):
    """random.sample / np.random.choice(replace=False) semantics of the replay draw: B distinct rows, every row
    equally likely (chi-square over many draws), a different draw per counter."""
    size, B = 1000, 64
    counts = np.zeros(size)
    for c in range(2000):
        idx = oracle.uniform_indices(3, c, size, B)
        assert len(set(idx.tolist())) == B and idx.min() >= 0 and idx.max() < size
        counts[idx] += 1
    expected = 2000 * B / size
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert abs(chi2 - size) < 6 * np.sqrt(2 * size), chi2                 # ~ chi-square with ~size degrees of freedom
    assert not np.array_equal(oracle.uniform_indices(3, 1, size, B), oracle.uniform_indices(3, 2, size, B))
