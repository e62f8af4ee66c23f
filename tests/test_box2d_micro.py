"""The env oracle against closed-form Box2D / gymnasium micro-scenarios (tests/box2d_micro.py; expectations in
tests/golden/box2d_micro.npz, derived in tests/golden/make_box2d_micro.py).  CPU; the HIP kernel runs the same scenarios
in tests/test_box2d_micro_gpu.py."""
import numpy as np

import box2d_micro as bm
from conftest import load_golden


def test_fixture_matches_its_derivation():
    g = load_golden("box2d_micro")
    m, inertia = bm.masses()
    assert np.array_equal(g["mass"], m) and np.array_equal(g["inertia"], inertia) and int(g["sleep_steps"]) == bm.sleep_steps()
    # hand values: hull area = 867 / 900 m^2 (trapezoid 34 x 10 + trapezoid (34 + 28) / 2 x 17, in pixels / 30^2) x density 5
    assert abs(m[0] - 5.0 * (34 * 10 + (34 + 28) / 2 * 17) / 900.0) < 1e-12
    assert abs(m[1] - (4 / 30) * (16 / 30)) < 1e-15 and abs(float(g["Mg_dt"]) - m.sum() * 0.2) < 1e-15


def test_oracle_mass_constants(oracle):
    g = load_golden("box2d_micro")
    c = oracle.lunar_constants().astype(np.float64)
    assert np.allclose(1.0 / c[:3], g["mass"], rtol=1e-6) and np.allclose(1.0 / c[3:], g["inertia"], rtol=1e-6)


def test_oracle_micro_scenarios(oracle):
    def make(n, seed):
        env = oracle.Env(oracle.LUNARLANDER, n, seed=seed)
        return env.reset, env.step, env.lunar_words
    checked = bm.run_scenarios(make, oracle.philox, load_golden("box2d_micro"))
    assert checked >= 16
