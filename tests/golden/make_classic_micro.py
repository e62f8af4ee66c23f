#!/usr/bin/env python3
"""Closed-form pins for CartPole-v1 and Pendulum-v1 (VERDICT r02 item 9 / SURVEY.md 8(c).2).

gymnasium is not installable here, so nothing can run its env code; what CAN be done independently of this repo's
restatement (oracle/gymrl_oracle.c, csrc/env_classic.hip) is to evaluate gymnasium's PUBLISHED one-step equations by
hand: this script does exactly that in plain Python float64 (`math` only — it imports neither the oracle nor the
package), from hand-picked states, and stores inputs + expected outputs as tests/golden/classic_micro.npz:

  CartPole-v1  (classic_control/cartpole.py: gravity 9.8, masscart 1.0, masspole 0.1, length 0.5 (half the pole),
               force_mag 10, tau 0.02, explicit Euler, theta threshold 12 * 2 * pi / 360, x threshold 2.4,
               reward 1.0 on every step including the terminating one, TimeLimit 500)
  Pendulum-v1  (classic_control/pendulum.py: max_speed 8, max_torque 2, dt 0.05, g 10, m 1, l 1;
               cost from the PRE-step state, new speed clipped, new angle from the clipped NEW speed, TimeLimit 200)

tests/classic_micro.py replays every scenario on the oracle (CPU suite) and on the HIP kernels (GPU suite).

    python tests/golden/make_classic_micro.py
"""
import math
import os

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ CartPole-v1, as published --------------------
def cartpole_step(state, action):
    x, x_dot, theta, theta_dot = state
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    total_mass = masspole + masscart
    polemass_length = masspole * length
    force = force_mag if action == 1 else -force_mag
    costheta, sintheta = math.cos(theta), math.sin(theta)
    temp = (force + polemass_length * theta_dot ** 2 * sintheta) / total_mass
    thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * costheta ** 2 / total_mass))
    xacc = temp - polemass_length * thetaacc * costheta / total_mass
    x = x + tau * x_dot                      # "euler": positions advance with the OLD velocities
    x_dot = x_dot + tau * xacc
    theta = theta + tau * theta_dot
    theta_dot = theta_dot + tau * thetaacc
    theta_threshold_radians, x_threshold = 12 * 2 * math.pi / 360, 2.4
    terminated = bool(x < -x_threshold or x > x_threshold or theta < -theta_threshold_radians or theta > theta_threshold_radians)
    return (x, x_dot, theta, theta_dot), 1.0, terminated


# ------------------------------------------------------------------ Pendulum-v1, as published --------------------
def angle_normalize(x):
    return ((x + math.pi) % (2 * math.pi)) - math.pi


def pendulum_step(state, u):
    th, thdot = state
    g, m, l, dt, max_speed, max_torque = 10.0, 1.0, 1.0, 0.05, 8.0, 2.0
    u = min(max(u, -max_torque), max_torque)
    costs = angle_normalize(th) ** 2 + 0.1 * thdot ** 2 + 0.001 * (u ** 2)
    newthdot = thdot + (3 * g / (2 * l) * math.sin(th) + 3.0 / (m * l ** 2) * u) * dt
    newthdot = min(max(newthdot, -max_speed), max_speed)
    newth = th + newthdot * dt
    return (newth, newthdot), -costs


def pendulum_obs(state):
    return (math.cos(state[0]), math.sin(state[0]), state[1])


def main():
    out = {}
    # ---- CartPole: one step from hand-picked states, both actions
    cp_states = [(0.0, 0.0, 0.0, 0.0), (0.01, -0.02, 0.03, 0.04), (-1.5, 0.7, -0.1, 1.2), (2.0, -1.0, 0.15, -2.0),
                 (0.3, 2.5, -0.2, 0.5), (-2.3, -0.4, 0.05, 3.0), (1.0, 0.0, 0.2, 0.0), (0.0, 1.0, -0.19, -0.8)]
    s0, a0, s1, t1 = [], [], [], []
    for st in cp_states:
        for a in (0, 1):
            nxt, r, term = cartpole_step(st, a)
            assert r == 1.0
            s0.append(st); a0.append(a); s1.append(nxt); t1.append(term)
    out.update(cp1_state=np.array(s0, np.float64), cp1_action=np.array(a0, np.int32), cp1_next=np.array(s1, np.float64),
               cp1_terminated=np.array(t1, np.uint8))
    # ---- CartPole: termination thresholds (strict inequalities), reward 1.0 on the terminating step
    lim = 12 * 2 * math.pi / 360
    edge = [(2.39, 0.6, 0.0, 0.0), (2.39, 0.4, 0.0, 0.0), (-2.39, -0.6, 0.0, 0.0), (-2.39, -0.4, 0.0, 0.0),
            (2.0, 20.0, 0.0, 0.0),                                    # lands on 2.0 + 0.02 * 20.0: whatever float64 says
            (0.0, 0.0, lim - 0.001, 0.1), (0.0, 0.0, lim - 0.003, 0.1), (0.0, 0.0, -(lim - 0.001), -0.1),
            (0.0, 0.0, -(lim - 0.003), -0.1), (0.0, 0.0, 0.2, 0.47), (0.0, 0.0, 0.2, 0.48)]
    es, ea, en, et = [], [], [], []
    for st in edge:
        for a in (0, 1):
            nxt, r, term = cartpole_step(st, a)
            es.append(st); ea.append(a); en.append(nxt); et.append(term)
    assert 0 < sum(et) < len(et)
    out.update(cpe_state=np.array(es, np.float64), cpe_action=np.array(ea, np.int32), cpe_next=np.array(en, np.float64),
               cpe_terminated=np.array(et, np.uint8))
    # ---- CartPole: 10-step trajectories (action patterns) incl. one that ends inside the window
    trajs = [((0.02, -0.01, 0.03, 0.02), [1, 0, 1, 1, 0, 0, 1, 0, 1, 0]), ((-0.04, 0.03, -0.02, -0.04), [0] * 10),
             ((0.0, 0.0, 0.1, 0.5), [1] * 10), ((1.0, 1.5, -0.05, 0.3), [1, 1, 0, 1, 1, 0, 1, 1, 0, 1])]
    ts, ta, tn, tt = [], [], [], []
    for st, acts in trajs:
        ts.append(st); ta.append(acts)
        row, term_row, alive = [], [], True
        for a in acts:
            if alive:
                st, r, term = cartpole_step(st, a)
                row.append(st); term_row.append(term)
                alive = not term
            else:                                   # after the episode ended the env resets: not pinned here
                row.append((math.nan,) * 4); term_row.append(False)
        tn.append(row); tt.append(term_row)
    assert any(any(r) for r in tt)
    out.update(cpt_state=np.array(ts, np.float64), cpt_actions=np.array(ta, np.int32), cpt_next=np.array(tn, np.float64),
               cpt_terminated=np.array(tt, np.uint8))
    # ---- Pendulum: one step, torques beyond the bound, speed clip, angle normalisation far from [-pi, pi)
    pd_states = [(0.0, 0.0), (math.pi / 2, 0.0), (-math.pi / 2, 1.0), (3.0, -0.5), (-3.1, 0.9), (math.pi, 0.0),
                 (3 * math.pi / 2, 2.0), (-7.0, -3.0), (0.5, 7.9), (-0.5, -7.95), (10.0, 8.0), (1.0, -8.0)]
    ps, pu, pn, pr = [], [], [], []
    for st in pd_states:
        for u in (-5.0, -2.0, -0.7, 0.0, 1.3, 2.0, 5.0):
            nxt, r = pendulum_step(st, float(np.float32(u)))        # the action arrives as float32
            ps.append(st); pu.append(u); pn.append(nxt); pr.append(r)
    out.update(pd1_state=np.array(ps, np.float64), pd1_u=np.array(pu, np.float32), pd1_next=np.array(pn, np.float64),
               pd1_reward=np.array(pr, np.float64),
               pd1_obs=np.array([pendulum_obs(s) for s in pn], np.float64))
    assert any(abs(n[1]) == 8.0 for n in pn)                       # the speed clip is exercised
    # ---- Pendulum: 10-step trajectories
    ptraj = [((math.pi - 0.1, 0.0), [2.0, -2.0, 1.0, 0.0, -1.0, 2.0, 2.0, -0.5, 0.5, 0.0]), ((0.1, 0.0), [0.0] * 10),
             ((-2.0, 0.5), [3.0] * 10), ((1.0, -1.0), [-2.0, -2.0, -2.0, 2.0, 2.0, 2.0, 0.3, -0.3, 1.7, -1.7])]
    qs, qu, qn, qr = [], [], [], []
    for st, us in ptraj:
        qs.append(st); qu.append(us)
        row, rr = [], []
        for u in us:
            st, r = pendulum_step(st, float(np.float32(u)))
            row.append(st); rr.append(r)
        qn.append(row); qr.append(rr)
    out.update(pdt_state=np.array(qs, np.float64), pdt_u=np.array(qu, np.float32), pdt_next=np.array(qn, np.float64),
               pdt_reward=np.array(qr, np.float64))
    out["time_limits"] = np.array([500, 200], np.int64)
    path = os.path.join(OUT, "classic_micro.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} B): {sum(et)} of {len(et)} edge cases terminate")


if __name__ == "__main__":
    main()
