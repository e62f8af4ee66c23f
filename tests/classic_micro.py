"""Closed-form scenarios for CartPole-v1 / Pendulum-v1, run on BOTH engines: the oracle (tests/test_classic_micro.py, CPU)
and the HIP kernels through the C-ABI (tests/test_classic_micro_gpu.py).  Expectations come from
tests/golden/classic_micro.npz — gymnasium's published equations evaluated by hand in float64 by
tests/golden/make_classic_micro.py, which shares no code with either engine.

An engine is an object with
    set_state(kind, states f64[n, k], ep_len)  -> None      (n envs put into the given states)
    step(actions)                              -> dict(obs, term_obs, rew, terminated, truncated, done, ep_len)
    reset(kind, n, seed)                       -> obs f32[n, D]
Tolerances: observations are the float32 cast of a float64 state; the engines may differ from the hand evaluation by an
ulp of float64 in sin / cos, so floats are compared at 1e-6 (relative form), flags and CartPole's reward exactly."""
import numpy as np

CARTPOLE, PENDULUM = 0, 1
TOL = 1e-6


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))))


def cartpole_one_step(eng, g):
    for pre in ("cp1", "cpe"):                       # hand-picked states, then the termination edges
        eng.set_state(CARTPOLE, g[pre + "_state"], 0)
        r = eng.step(g[pre + "_action"])
        want = g[pre + "_next"].astype(np.float32)
        assert _close(r["term_obs"], want), pre       # the pre-reset observation of the step
        assert np.array_equal(r["terminated"], g[pre + "_terminated"]), pre
        assert np.all(r["truncated"] == 0) and np.array_equal(r["done"], g[pre + "_terminated"])
        assert np.all(r["rew"] == 1.0)                # 1.0 on every step, the terminating one included
        alive = g[pre + "_terminated"] == 0
        assert _close(r["obs"][alive], want[alive])   # no reset where the episode goes on
        assert np.all(np.abs(r["obs"][~alive]) <= 0.05)      # reset observation where it ended: U(-0.05, 0.05)^4
        assert np.array_equal(r["ep_len"][~alive], np.ones(int((~alive).sum()), np.int32))


def cartpole_trajectories(eng, g):
    eng.set_state(CARTPOLE, g["cpt_state"], 0)
    acts, want, term = g["cpt_actions"], g["cpt_next"], g["cpt_terminated"]
    live = np.ones(len(acts), bool)
    for t in range(acts.shape[1]):
        r = eng.step(acts[:, t])
        assert _close(r["term_obs"][live], want[live, t].astype(np.float32), 2e-6), t
        assert np.array_equal(r["terminated"][live], term[live, t]), t
        live &= term[:, t] == 0
    assert not live.all()                            # at least one trajectory ended inside the window


def cartpole_time_limit(eng, g):
    limit = int(g["time_limits"][0])
    st = np.tile(np.array([[0.0, 0.0, 0.0, 0.0]]), (4, 1))
    eng.set_state(CARTPOLE, st, [limit - 2, limit - 2, limit - 3, 0])
    r = eng.step(np.array([1, 0, 1, 0], np.int32))   # step limit-1 of envs 0, 1
    assert np.all(r["truncated"] == 0) and np.all(r["done"] == 0)
    r = eng.step(np.array([0, 1, 0, 1], np.int32))   # step `limit` of envs 0, 1: truncated, not terminated
    assert list(r["truncated"]) == [1, 1, 0, 0] and np.all(r["terminated"] == 0) and list(r["done"]) == [1, 1, 0, 0]
    assert list(r["ep_len"][:2]) == [limit, limit] and np.all(r["rew"] == 1.0)
    r = eng.step(np.array([1, 0, 1, 0], np.int32))
    assert list(r["truncated"]) == [0, 0, 1, 0]


def pendulum_one_step(eng, g):
    eng.set_state(PENDULUM, g["pd1_state"], 0)
    r = eng.step(g["pd1_u"].reshape(-1, 1))
    assert _close(r["obs"], g["pd1_obs"].astype(np.float32))
    assert _close(r["rew"], g["pd1_reward"].astype(np.float32))       # minus the cost of the PRE-step state
    assert np.all(r["terminated"] == 0) and np.all(r["truncated"] == 0)
    assert np.all(np.abs(r["obs"][:, 2]) <= 8.0) and np.any(np.abs(r["obs"][:, 2]) == 8.0)     # speed clip reached


def pendulum_trajectories(eng, g):
    eng.set_state(PENDULUM, g["pdt_state"], 0)
    us, want, rew = g["pdt_u"], g["pdt_next"], g["pdt_reward"]
    for t in range(us.shape[1]):
        r = eng.step(us[:, t].reshape(-1, 1))
        obs = np.stack([np.cos(want[:, t, 0]), np.sin(want[:, t, 0]), want[:, t, 1]], 1).astype(np.float32)
        assert _close(r["obs"], obs, 2e-6), t
        assert _close(r["rew"], rew[:, t].astype(np.float32), 2e-6), t


def pendulum_time_limit(eng, g):
    limit = int(g["time_limits"][1])
    eng.set_state(PENDULUM, np.array([[0.1, 0.0], [3.0, 1.0]]), [limit - 2, 0])
    u = np.zeros((2, 1), np.float32)
    r = eng.step(u)
    assert np.all(r["done"] == 0)
    r = eng.step(u)
    assert list(r["truncated"]) == [1, 0] and np.all(r["terminated"] == 0) and int(r["ep_len"][0]) == limit
    assert abs(float(r["obs"][0, 2])) <= 1.0          # env 0 restarted: speed ~ U(-1, 1)


def reset_ranges(eng, g):
    o = eng.reset(CARTPOLE, 4096, seed=11)
    assert o.shape == (4096, 4) and np.all(np.abs(o) <= 0.05)
    assert np.all(np.abs(o.mean(0)) < 0.003) and np.all(np.abs(o.std(0) - 0.1 / np.sqrt(12)) < 0.002)    # uniform
    assert o.min() < -0.049 and o.max() > 0.049
    o = eng.reset(PENDULUM, 4096, seed=12)
    th = np.arctan2(o[:, 1], o[:, 0])
    assert _close(o[:, 0] ** 2 + o[:, 1] ** 2, 1.0) and np.all(np.abs(o[:, 2]) <= 1.0)
    assert abs(th.mean()) < 0.12 and abs(th.std() - 2 * np.pi / np.sqrt(12)) < 0.06 and th.min() < -3.1 and th.max() > 3.1
    assert abs(o[:, 2].mean()) < 0.04 and abs(o[:, 2].std() - 2 / np.sqrt(12)) < 0.02


SCENARIOS = [cartpole_one_step, cartpole_trajectories, cartpole_time_limit, pendulum_one_step, pendulum_trajectories,
             pendulum_time_limit, reset_ranges]
