"""The closed-form CartPole-v1 / Pendulum-v1 scenarios (tests/classic_micro.py) on the HIP steppers, through the C-ABI
(gymrl_env_reset / gymrl_env_step).  States are written into the caller-owned state buffer in the layout
include/gymrl.h documents for the classic envs (SoA fields, each padded to 256 bytes)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import classic_micro as cm            # noqa: E402
from conftest import load_golden      # noqa: E402


class HipEngine:
    def __init__(self):
        from gymrl_amd import ops
        self.ops, self.dev = ops, torch.device("cuda:0")

    def _fields(self, n, k):
        """Byte offsets of the k float64 state fields, ep_ret, ep_len, episode (each field rounded up to 256 B)."""
        pad = lambda b: (b + 255) // 256 * 256          # noqa: E731
        offs, off = [], 0
        for size in [8] * (k + 1) + [4, 4]:
            offs.append(off)
            off += pad(size * n)
        return offs, off

    def set_state(self, kind, states, ep_len):
        ops = self.ops
        states = np.asarray(states, np.float64)
        n, k = states.shape
        self.kind = ops.CARTPOLE if kind == cm.CARTPOLE else ops.PENDULUM
        self.n, self.seed = n, 5
        self.state = ops.env_state(self.kind, n, self.dev)
        obs = torch.empty(n, 4 if kind == cm.CARTPOLE else 3, device=self.dev)
        ops.env_reset(self.kind, self.state, n, self.seed, 0, obs)
        offs, total = self._fields(n, k)
        assert total == self.state.numel()               # the documented layout is the whole buffer
        for j in range(k):
            self.state[offs[j]:offs[j] + 8 * n].view(torch.float64).copy_(torch.from_numpy(states[:, j].copy()))
        self.state[offs[k]:offs[k] + 8 * n].view(torch.float64).zero_()
        lens = np.broadcast_to(np.asarray(ep_len, np.int32), (n,)).copy()
        self.state[offs[k + 1]:offs[k + 1] + 4 * n].view(torch.int32).copy_(torch.from_numpy(lens))

    def step(self, actions):
        ops, n, dev = self.ops, self.n, self.dev
        D = 4 if self.kind == ops.CARTPOLE else 3
        act = torch.from_numpy(np.ascontiguousarray(actions, np.int32 if self.kind == ops.CARTPOLE else np.float32)).to(dev)
        obs, tobs = torch.empty(n, D, device=dev), torch.empty(n, D, device=dev)
        rew = torch.empty(n, device=dev)
        term, trunc, done = (torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(3))
        ep_ret, ep_len = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        ops.env_step(self.kind, self.state, n, self.seed, 0, act, obs, rew, term, trunc, term_obs_out=tobs, done_out=done,
                     ep_ret_out=ep_ret, ep_len_out=ep_len)
        c = lambda t: t.cpu().numpy()                      # noqa: E731
        return dict(obs=c(obs), term_obs=c(tobs), rew=c(rew), terminated=c(term), truncated=c(trunc), done=c(done),
                    ep_ret=c(ep_ret), ep_len=c(ep_len))

    def reset(self, kind, n, seed):
        ops = self.ops
        k = ops.CARTPOLE if kind == cm.CARTPOLE else ops.PENDULUM
        state = ops.env_state(k, n, self.dev)
        obs = torch.empty(n, 4 if kind == cm.CARTPOLE else 3, device=self.dev)
        ops.env_reset(k, state, n, seed, 0, obs)
        return obs.cpu().numpy()


@pytest.mark.parametrize("scenario", cm.SCENARIOS, ids=lambda f: f.__name__)
def test_hip_steppers_match_the_published_equations(scenario):
    scenario(HipEngine(), load_golden("classic_micro"))


def test_hip_reset_equals_oracle_reset(oracle):
    """(the scenarios above compare both engines with the hand evaluation; resets are keyed Philox draws, compared directly)"""
    eng = HipEngine()
    for kind, ok in ((cm.CARTPOLE, oracle.CARTPOLE), (cm.PENDULUM, oracle.PENDULUM)):
        assert np.array_equal(eng.reset(kind, 777, seed=9), oracle.Env(ok, 777, seed=9).reset())
