"""Pin the CPU oracle against golden vectors captured from the reference's own
Python (tests/golden/make_golden.py).  CPU-only; runs in the build container."""
import numpy as np
import pytest

from conftest import cov_clip_mul, load_golden, rel_close

TOL = 1e-5  # BASELINE.json north_star: returns/advantages within 1e-5 (relative form)


def test_gae_g1_matches_reference(oracle):
    g = load_golden("gae_g1")
    for k in range(int(g["n_cases"])):
        rew, val, done = g[f"c{k}_rew"][:, None], g[f"c{k}_val"][:, None], g[f"c{k}_done"][:, None]
        nv = np.array([g[f"c{k}_next"]], np.float32)
        adv, ret, mom = oracle.gae(rew, val, done, nv, float(g["gamma"]), float(g["lam"]), want_moments=True)
        # the f64 recursion is bit-identical to numpy's; only the f32 store rounds
        assert np.array_equal(adv[:, 0], g[f"c{k}_adv"].astype(np.float32)), k
        assert np.array_equal(ret[:, 0], g[f"c{k}_ret"].astype(np.float32)), k
        assert rel_close(adv[:, 0], g[f"c{k}_adv"]) <= TOL
        if rew.shape[0] > 1 and g[f"c{k}_adv"].std() > 0:
            # P5: whole-rollout normalisation, population std (ppo_lunarlander.py:236)
            norm = oracle.normalize(adv, mom, ddof=0, eps=1e-8)
            assert rel_close(norm[:, 0], g[f"c{k}_norm"]) <= TOL, k


def test_gae_g2_matches_reference(oracle):
    g = load_golden("gae_g2")
    for k in range(int(g["n_cases"])):
        adv, vt, mom = oracle.gae_dw(g[f"c{k}_rew"], g[f"c{k}_val"], g[f"c{k}_nval"], g[f"c{k}_done"],
                                     g[f"c{k}_dw"], float(g["gamma"]), float(g["lam"]))
        assert np.array_equal(vt, g[f"c{k}_vt"]), k           # float32 recursion: bit-exact
        advn = oracle.normalize(adv, mom, ddof=1, eps=1e-8)   # torch.std is unbiased (utils/buffer.py:33)
        assert rel_close(advn, g[f"c{k}_advn"]) <= TOL, k


def test_gae_g3_matches_reference(oracle):
    g = load_golden("gae_g3")
    for k in range(int(g["n_cases"])):
        adv, ret = oracle.gae_decoupled(g[f"c{k}_rew"][:, None], g[f"c{k}_val"][:, None], g[f"c{k}_done"][:, None],
                                        np.array([g[f"c{k}_next"]], np.float32), float(g["gamma"]),
                                        float(g["lam_actor"]), float(g["lam_critic"]))
        assert np.array_equal(adv[:, 0], g[f"c{k}_adv"].astype(np.float32))
        assert np.array_equal(ret[:, 0], g[f"c{k}_ret"].astype(np.float32))


def test_categorical_matches_torch(oracle):
    g = load_golden("categorical")
    act, logp, ent, _ = oracle.categorical_sample(g["logits"], noise_exp=g["noise_exp"])
    assert np.array_equal(act, g["action"])          # integer draws bit-exact
    assert rel_close(logp, g["logp"]) <= TOL
    assert rel_close(ent, g["entropy"]) <= TOL
    act_d, _, _, _ = oracle.categorical_sample(g["logits"], deterministic=True)
    assert np.array_equal(act_d, g["argmax"])


def test_det_math_accuracy(oracle):
    x = np.linspace(-80, 80, 4001).astype(np.float32)
    e = oracle.expf(x)
    assert np.max(np.abs(e / np.exp(x.astype(np.float64)) - 1)) < 3e-7
    y = np.exp(np.linspace(-30, 30, 4001)).astype(np.float32)
    assert np.max(np.abs(oracle.logf(y) - np.log(y.astype(np.float64)))) < 2e-6
    t = np.linspace(-12, 12, 2001).astype(np.float32)
    assert np.max(np.abs(oracle.tanhf(t) - np.tanh(t.astype(np.float64)))) < 3e-7
    s, c = oracle.sincosf(np.linspace(-50, 50, 4001).astype(np.float32))
    a = np.linspace(-50, 50, 4001).astype(np.float32).astype(np.float64)
    assert np.max(np.abs(s - np.sin(a))) < 3e-7 and np.max(np.abs(c - np.cos(a))) < 3e-7


def test_ppo_loss_matches_autograd(oracle):
    g = load_golden("ppo_loss")
    cfg = tuple(float(x) for x in g["cfg"])
    for k in range(int(g["n_cases"])):
        dl, dv, met = oracle.ppo_loss_fwd_bwd(g[f"c{k}_logits"], g[f"c{k}_values"], g[f"c{k}_actions"],
                                              g[f"c{k}_old_lp"], g[f"c{k}_adv"], g[f"c{k}_ret"], cfg)
        B = g[f"c{k}_logits"].shape[0]
        scale = np.abs(g[f"c{k}_dlogits"]).max()
        assert np.max(np.abs(dl - g[f"c{k}_dlogits"])) <= TOL * scale
        assert np.max(np.abs(dv - g[f"c{k}_dvalues"])) <= TOL * np.abs(g[f"c{k}_dvalues"]).max()
        assert rel_close(met / B, g[f"c{k}_metrics"], 1e-5) <= 1e-5


def test_ppo_full_loss_matches_autograd(oracle):
    g = load_golden("ppo_full_loss")
    cfg = tuple(float(x) for x in g["cfg"])
    for k in range(int(g["n_cases"])):
        dl, dv, met = oracle.ppo_full_loss_fwd_bwd(g[f"c{k}_logits"], g[f"c{k}_values"], g[f"c{k}_actions"],
                                                   g[f"c{k}_old_lp"], g[f"c{k}_old_ent"], g[f"c{k}_adv"],
                                                   g[f"c{k}_ret"], cfg, corr_mul=cov_clip_mul(g, k))
        B = g[f"c{k}_logits"].shape[0]
        assert np.max(np.abs(dl - g[f"c{k}_dlogits"])) <= TOL * np.abs(g[f"c{k}_dlogits"]).max()
        assert np.max(np.abs(dv - g[f"c{k}_dvalues"])) <= TOL * np.abs(g[f"c{k}_dvalues"]).max()
        ref = g[f"c{k}_metrics"]
        got = np.array([met[0] / B, met[1] / B, met[2] / B, met[3] / B, met[4] / B, met[5] / B,
                        (met[8] - met[6] * met[7] / B) / B])
        assert rel_close(got, ref, 1e-5) <= 1e-5


def test_adam_matches_torch(oracle):
    g = load_golden("ppo_loss")
    lr, b1, b2, eps, max_norm = (float(x) for x in g["adam"])
    for k in range(int(g["n_cases"])):
        n = g[f"c{k}_grads"].size
        p, _, m, v = oracle.adam_step(g[f"c{k}_params0"], g[f"c{k}_grads"], np.zeros(n), np.zeros(n), lr, b1, b2,
                                      eps, 1, max_grad_norm=max_norm)
        assert abs(np.sqrt(oracle.sqnorm(g[f"c{k}_grads"])[0]) - float(g[f"c{k}_total_norm"])) <= 1e-5 * float(g[f"c{k}_total_norm"])
        assert np.max(np.abs(p - g[f"c{k}_params1"])) <= 1e-6
        assert rel_close(m, g[f"c{k}_m1"], 1e-5) <= 1e-5 and rel_close(v, g[f"c{k}_v1"], 1e-5) <= 1e-5
    a = load_golden("adam")
    for k in range(int(a["n_cases"])):
        lr, b1, b2, eps, max_norm, clamp = (float(x) for x in a[f"c{k}_hp"])
        p = a[f"c{k}_p0"].copy()
        m, v = np.zeros_like(p), np.zeros_like(p)
        for step, gr in enumerate(a[f"c{k}_grads"], 1):
            p, _, m, v = oracle.adam_step(p, gr, m, v, lr, b1, b2, eps, step, max_grad_norm=max_norm, clamp_abs=clamp)
        assert np.max(np.abs(p - a[f"c{k}_p5"])) <= 2e-6, k
        assert rel_close(m, a[f"c{k}_m5"], 1e-5) <= 1e-5 and rel_close(v, a[f"c{k}_v5"], 1e-5) <= 1e-5


def test_soft_update_matches_torch(oracle):
    g = load_golden("soft_update")
    out = oracle.soft_update(g["target"], g["source"], float(g["tau"]))
    assert np.max(np.abs(out - g["out"])) <= 1e-7


def test_ppo_trace_fixture_is_self_consistent(oracle):
    """Row H1 fixture (reference train() on the scripted env): the scripted env replays the recorded
    observation / reward / done sequence from the recorded actions; the oracle's categorical rule on the
    initial network's logits + the recorded Exp(1) draws reproduces rollout 0's actions / log-probs /
    values; the oracle GAE reproduces the recorded adv/ret."""
    from scripted_env import ScriptedEnv
    g = load_golden("ppo_trace")
    env = ScriptedEnv(8, 4)
    for r in range(2):
        p = f"r{r}_"
        obs, _ = env.reset()
        ep_ret, finished = 0.0, []
        for t in range(int(g["cfg"][0])):
            assert np.array_equal(obs, g[p + "states"][t]), (r, t)
            obs, rew, te, tr, _ = env.step(int(g[p + "actions"][t]))
            assert rew == g[p + "rewards"][t] and int(te or tr) == g[p + "dones"][t]
            ep_ret += rew
            if te or tr:
                finished.append(ep_ret)
                ep_ret = 0.0
                obs, _ = env.reset()
        prev = [] if r == 0 else list(g["r0_episode_rewards"])
        assert np.allclose((prev + finished)[-100:], g[p + "episode_rewards"])
        adv, ret, _ = oracle.gae(g[p + "rewards"].astype(np.float32)[:, None], g[p + "values"][:, None],
                                 g[p + "dones"][:, None], np.array([g[p + "next_value"]], np.float32),
                                 0.99, 0.95, want_moments=True)
        assert np.array_equal(adv[:, 0], g[p + "adv"].astype(np.float32))
        assert np.array_equal(ret[:, 0], g[p + "ret"].astype(np.float32))
    # P2 on rollout 0: numpy forward of the recorded initial ActorCritic (:63-90)
    def lin(x, name):
        return x @ g["init_" + name + ".weight"].T.astype(np.float64) + g["init_" + name + ".bias"]
    x = g["r0_states"].astype(np.float64)
    h = np.tanh(lin(np.tanh(lin(x, "shared.0")), "shared.2"))
    logits = lin(np.tanh(lin(h, "actor.0")), "actor.2").astype(np.float32)
    value = lin(np.tanh(lin(h, "critic.0")), "critic.2")[:, 0]
    act, logp, _, _ = oracle.categorical_sample(logits, noise_exp=g["noise_exp"][0, :, 0])
    assert np.array_equal(act, g["r0_actions"])
    assert rel_close(logp, g["r0_log_probs"]) <= TOL and rel_close(value, g["r0_values"]) <= TOL
    # O2: lr = lr0 * (1 - step_count / max_train_steps) before each rollout (:337-341)
    assert np.allclose(g["lr"], [float(g["lr0"]), float(g["lr0"]) * 0.5])


def test_permutation_is_a_keyed_bijection(oracle):
    """P6 epoch shuffle restatement: a bijection of [0, M) for every M (powers of two, odd sizes, tiny sizes),
    a different one per counter, with no visible structure (fixed points ~ 1, |perm[i+1] - perm[i]| ~ M/3)."""
    for M in (1, 2, 3, 5, 64, 100, 4097, 100003, 1 << 18):
        p = oracle.permutation(11, 5, M)
        assert np.array_equal(np.sort(p), np.arange(M, dtype=np.int32)), M
    M = 1 << 18
    p1, p2 = oracle.permutation(11, 5, M), oracle.permutation(11, 6, M)
    assert np.array_equal(p1, oracle.permutation(11, 5, M))
    assert (p1 == p2).sum() < 20 and (p1 == np.arange(M)).sum() < 20
    assert abs(np.abs(np.diff(p1.astype(np.int64))).mean() / M - 1 / 3) < 0.01
    assert abs(np.corrcoef(p1, np.arange(M))[0, 1]) < 0.01
    # every minibatch slice sees the rollout uniformly: slice means within 4 sigma of (M - 1) / 2
    mb = p1.reshape(32, -1).astype(np.float64).mean(1)
    sigma = M / np.sqrt(12.0 * (M // 32))
    assert np.all(np.abs(mb - (M - 1) / 2) < 4 * sigma)


def test_ppo_full_cov_clip_branch_matches_reference(oracle):
    """L3 with clip_cov_ratio > 0 (ppo_full_lunarlander.py:594-616): the rows gymrl_amd's cov_clip_mask takes out,
    fed the reference's own randperm, are exactly the rows the reference zeroed (golden case 2); the gradients and
    metrics with that mask are covered by test_ppo_full_loss_matches_autograd."""
    g = load_golden("ppo_full_loss")
    mul = cov_clip_mul(g, 2)
    ratio = float(g["c2_cov_cfg"][0])
    assert (mul == 0).sum() == max(int(len(g["c2_cov_perm"]) * ratio), 1)
    assert np.all(g["c2_corr"][mul == 0] == 0)                               # every picked row is out of the means
    assert cov_clip_mul(g, 0) is None and cov_clip_mul(g, 1) is None         # default ratio 0: branch dead
