"""CPU checks of the oracle's env restatements (gymnasium semantics; parity unpinned —
these pin the documented behaviour the HIP kernels are then compared against)."""
import numpy as np

from test_hip_parity import _lander_heuristic


def test_cartpole_semantics(oracle):
    env = oracle.Env(oracle.CARTPOLE, 64, seed=3)
    o = env.reset()
    assert o.shape == (64, 4) and np.all(np.abs(o) <= 0.05)
    lens = []
    for _ in range(600):
        r = env.step(np.ones(64, np.int32))                 # always push right: falls in ~10 steps
        assert np.all(r["rew"] == 1.0)
        d = r["done"].astype(bool)
        lens += list(r["ep_len"][d])
        assert np.all(np.abs(r["obs"][d]) <= 0.05)           # post-reset observation where done
        t = r["terminated"].astype(bool)
        assert np.all((np.abs(r["term_obs"][t, 0]) > 2.4) | (np.abs(r["term_obs"][t, 2]) > 12 * np.pi / 180))
    assert 7 <= np.mean(lens) <= 12


def test_pendulum_semantics(oracle):
    env = oracle.Env(oracle.PENDULUM, 32, seed=4)
    o = env.reset()
    assert np.allclose(o[:, 0] ** 2 + o[:, 1] ** 2, 1.0, atol=1e-6) and np.all(np.abs(o[:, 2]) <= 1.0)
    for s in range(1, 401):
        r = env.step(np.full((32, 1), 5.0, np.float32))      # clipped to the 2.0 torque bound
        assert np.all(r["terminated"] == 0)
        assert np.all(r["truncated"] == (1 if s % 200 == 0 else 0))
        assert np.all(r["rew"] <= 0.0) and np.all(r["rew"] >= -(np.pi ** 2 + 0.1 * 64 + 0.001 * 4) - 1e-4)
        assert np.all(np.abs(r["term_obs"][:, 2]) <= 8.0)


def test_lunarlander_reset_distribution_and_determinism(oracle):
    a = oracle.Env(oracle.LUNARLANDER, 512, seed=5).reset()
    b = oracle.Env(oracle.LUNARLANDER, 512, seed=5).reset()
    assert np.array_equal(a, b)
    c = oracle.Env(oracle.LUNARLANDER, 256, seed=5, env_id0=256).reset()
    assert np.array_equal(a[256:], c)                         # streams are keyed by GLOBAL env id (multi-GPU sharding)
    # gymnasium's LunarLander-v3 reset: x ~ 0, y ~ 1.40..1.42, |vx| < 0.83, legs up
    assert np.all(np.abs(a[:, 0]) < 0.01) and np.all((a[:, 1] > 1.39) & (a[:, 1] < 1.43))
    assert 0.3 < a[:, 2].std() < 0.6 and np.all(np.abs(a[:, 2]) < 0.85)
    assert np.all(a[:, 6:] == 0)


def test_lunarlander_heuristic_lands(oracle):
    n = 96
    env = oracle.Env(oracle.LUNARLANDER, n, seed=7)
    o = env.reset()
    rets, seen = [], np.zeros(n, bool)
    landed = 0
    for _ in range(1001):
        r = env.step(_lander_heuristic(o))
        o = r["obs"]
        d = r["done"].astype(bool) & ~seen
        rets += list(r["ep_ret"][d])
        landed += int((r["rew"][d] == 100).sum())
        seen |= r["done"].astype(bool)
        if seen.all():
            break
    assert seen.all()
    assert np.mean(rets) > 150 and landed >= 0.7 * n          # gymnasium's heuristic scores ~200+


def test_lunarlander_random_policy_statistics(oracle):
    n = 64
    env = oracle.Env(oracle.LUNARLANDER, n, seed=1)
    env.reset()
    rng = np.random.default_rng(0)
    rets, lens = [], []
    for _ in range(400):
        r = env.step(rng.integers(0, 4, size=n).astype(np.int32))
        assert np.isfinite(r["obs"]).all()
        d = r["done"].astype(bool)
        rets += list(r["ep_ret"][d])
        lens += list(r["ep_len"][d])
    # a uniform-random policy on gymnasium's LunarLander: about -180 +- 100 over ~90-110 steps
    assert -260 < np.mean(rets) < -120 and 70 < np.mean(lens) < 130
