"""The closed-form CartPole-v1 / Pendulum-v1 scenarios (tests/classic_micro.py) on the oracle: the published equations,
evaluated by hand in the fixture generator, pin the restatement the HIP kernels are compared against."""
import numpy as np
import pytest

import classic_micro as cm
from conftest import load_golden


class OracleEngine:
    def __init__(self, orc):
        self.orc, self.env = orc, None

    def set_state(self, kind, states, ep_len):
        states = np.asarray(states, np.float64)
        self.env = self.orc.Env(self.orc.CARTPOLE if kind == cm.CARTPOLE else self.orc.PENDULUM, len(states), seed=5)
        self.env.reset()
        self.env.set_classic_state(states, ep_len)

    def step(self, actions):
        return self.env.step(actions)

    def reset(self, kind, n, seed):
        return self.orc.Env(self.orc.CARTPOLE if kind == cm.CARTPOLE else self.orc.PENDULUM, n, seed=seed).reset()


@pytest.mark.parametrize("scenario", cm.SCENARIOS, ids=lambda f: f.__name__)
def test_oracle_matches_the_published_equations(oracle, scenario):
    scenario(OracleEngine(oracle), load_golden("classic_micro"))
