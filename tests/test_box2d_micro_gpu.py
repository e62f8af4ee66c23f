"""The HIP LunarLander stepper (through the C-ABI) against the closed-form Box2D / gymnasium micro-scenarios of
tests/box2d_micro.py — the same checks tests/test_box2d_micro.py runs on the oracle — and word for word against the oracle
on the scenario states (the whole 144-word world: bodies, sleep timers, joint impulses, manifolds, terrain, flags)."""
import numpy as np
import pytest

import box2d_micro as bm
from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _hip(n, seed):
    from gymrl_amd import ops
    dev = torch.device("cuda:0")
    state = ops.env_state(ops.LUNARLANDER, n, dev)
    obs, tobs = torch.empty(n, 8, device=dev), torch.empty(n, 8, device=dev)
    rew = torch.empty(n, device=dev)
    term, trunc, done = (torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(3))

    def reset():
        ops.env_reset(ops.LUNARLANDER, state, n, seed, 0, obs)
        return obs.cpu().numpy()

    def step(action):
        ops.env_step(ops.LUNARLANDER, state, n, seed, 0, torch.from_numpy(np.asarray(action, np.int32)).to(dev), obs, rew,
                     term, trunc, term_obs_out=tobs, done_out=done)
        return dict(obs=obs.cpu().numpy(), rew=rew.cpu().numpy(), done=done.cpu().numpy(), terminated=term.cpu().numpy())

    def words():        # the first field of the state buffer: u32[144][n] (env_lunar_device.hpp LunarState)
        return state[:144 * n * 4].view(torch.int32).view(144, n).cpu().numpy().view(np.uint32)
    return reset, step, words


def test_hip_micro_scenarios(oracle):
    checked = bm.run_scenarios(_hip, oracle.philox, load_golden("box2d_micro"))
    assert checked >= 16


def test_hip_world_words_equal_oracle_on_a_landing(oracle):
    """Every word of every world after every step of a heuristic landing (contacts, warm-start impulses, sleep timers,
    joint state) — not only the observations — is bit-identical between the HIP kernel and the oracle."""
    n, seed = 32, 18
    reset, step, words = _hip(n, seed)
    env = oracle.Env(oracle.LUNARLANDER, n, seed=seed)
    o = reset()
    assert np.array_equal(o, env.reset()) and np.array_equal(words(), env.lunar_words())
    contacts = 0
    for t in range(400):
        a = bm.heuristic(o)
        r, q = step(a), env.step(a)
        assert np.array_equal(r["obs"], q["obs"]) and np.array_equal(r["rew"], q["rew"]) and np.array_equal(r["done"], q["done"])
        w, v = words(), env.lunar_words()
        assert np.array_equal(w, v), (t, np.argwhere(w != v)[:5])
        contacts += int(bm.World(w).contacts.sum())
        o = r["obs"]
    assert contacts > 1000
