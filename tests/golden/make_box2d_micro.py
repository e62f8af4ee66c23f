#!/usr/bin/env python3
"""Closed-form expectations for the LunarLander micro-scenarios (tests/box2d_micro.py), written to box2d_micro.npz.

No reference code is involved (gymnasium / Box2D are third-party and absent): every number is derived here from the
published shapes and constants of gymnasium's LunarLander-v3 and Box2D 2.3's documented solver rules.

  mass[3], inertia[3]   hull: LANDER_POLY / 30 at density 5 — polygon area A = 1/2 sum (x_i y_{i+1} - x_{i+1} y_i),
                        mass = 5 A; second moment about the origin 5/12 sum (x_i y_{i+1} - x_{i+1} y_i)(x_i^2 + x_i x_{i+1}
                        + x_{i+1}^2 + y_i^2 + y_i y_{i+1} + y_{i+1}^2), shifted to the centroid (parallel axes).
                        legs: boxes 4/30 x 16/30 at density 1: m = w h, I = m (w^2 + h^2) / 12.
  Mg_dt                 (m_hull + 2 m_leg) * 10 m/s^2 * (1/50 s): what gravity adds to the total momentum per step and what
                        the contact impulses of a lander at rest must add up to.
  sleep_steps           consecutive quiet steps until b2Island puts the island to sleep: the smallest k with
                        fl32(k additions of fl32(1/50)) >= 0.5f.
  joint_lower/upper     revolute joint limits of the legs, gymnasium: i = -1: [+0.9 - 0.5, +0.9], i = +1: [-0.9, -0.9 + 0.5].
  angular_slop          b2_angularSlop = 2 degrees.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import box2d_micro as bm  # noqa: E402

m, inertia = bm.masses()
out = dict(mass=m, inertia=inertia, Mg_dt=np.float64(m.sum() * bm.G * bm.DT), sleep_steps=np.int64(bm.sleep_steps()),
           joint_lower=np.array([0.9 - 0.5, -0.9]), joint_upper=np.array([0.9, -0.9 + 0.5]),
           angular_slop=np.float64(2.0 / 180.0 * np.pi))
np.savez_compressed(os.path.join(HERE, "box2d_micro.npz"), **out)
print({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in out.items()})
