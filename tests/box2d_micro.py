"""Closed-form micro-scenarios for the LunarLander stepper (shared by the CPU test of the oracle and the GPU test of
the HIP kernel).  gymnasium / Box2D are not installable here, so the env rows cannot be pinned bit for bit; what CAN
be pinned without them is physics that does not depend on the restatement:

  * the mass and rotational inertia of the hull polygon and the leg boxes (shoelace / polygon second-moment formulas
    on gymnasium's published vertex lists and densities);
  * free flight: a sequential-impulse solver exchanges equal and opposite impulses, so the total linear momentum of the
    three bodies changes by exactly M g dt per step (semi-implicit Euler) whatever the joints do;
  * engines: the change of total momentum of a step with the main / a side engine firing is gymnasium's impulse
    formula for that step's two dispersion draws, minus M g dt;
  * revolute joints: after the legs have swung open the joint angle sits inside [lower, upper] (+- angular slop);
  * sleep: Box2D adds dt to a body's sleep time on every step it spends under the velocity tolerances and puts the
    island to sleep once the minimum reaches 0.5 s — 0.02f accumulated in float32 crosses 0.5f at the 25th addition
    or the 26th, computed here in float32; the env ends with +100 at exactly that step of the quiet run;
  * rest on the flat helipad: the accumulated normal impulses of the contact points add up to M g dt.

Everything takes worlds in the 144-word layout of the HIP state buffer (env_lunar_device.hpp world_io); the oracle
exports the same layout (orc_lunar_get_words).  Expected numbers live in tests/golden/box2d_micro.npz
(tests/golden/make_box2d_micro.py holds the derivations).
"""
import numpy as np

SCALE, FPS = 30.0, 50.0
DT = 1.0 / FPS
G = 10.0
MAIN_ENGINE_POWER, SIDE_ENGINE_POWER = 13.0, 0.6
SIDE_ENGINE_HEIGHT, SIDE_ENGINE_AWAY = 14.0, 12.0
LANDER_POLY = [(-14, +17), (-17, 0), (-17, -10), (+17, -10), (+17, 0), (+14, +17)]
LEG_W, LEG_H = 2, 8
RNG_ENV_STEP = 0x20000000


def f32(words, k):
    return np.ascontiguousarray(words[k]).view(np.float32)


class World:
    """Decoded view of u32[144, n] world words."""

    def __init__(self, words):
        w = np.asarray(words, np.uint32)
        self.n = w.shape[1]
        b = np.stack([f32(w, k) for k in range(18)]).reshape(3, 6, self.n)
        self.c = b[:, 0:2].astype(np.float64)            # [body, xy, n]
        self.a = b[:, 2].astype(np.float64)
        self.v = b[:, 3:5].astype(np.float64)
        self.w = b[:, 5].astype(np.float64)
        self.sleep = np.stack([f32(w, 18 + k) for k in range(3)])
        mf = w[31:31 + 96].reshape(3, 2, 16, self.n)
        self.mf_count = mf[:, :, 0].astype(np.int64)
        ni = np.stack([np.ascontiguousarray(mf[:, :, 9]).view(np.float32), np.ascontiguousarray(mf[:, :, 14]).view(np.float32)], 2)
        valid = np.arange(2)[None, None, :, None] < self.mf_count[:, :, None, :]
        self.normal_impulse_sum = np.where(valid, ni, 0.0).sum(axis=(0, 1, 2))
        self.contacts = self.mf_count.sum(axis=(0, 1))
        self.flags = w[142].astype(np.int64)


def polygon_mass_inertia(verts, density):
    """Area, centroid and second moment about the centroid of a simple polygon (shoelace formulas), in metres."""
    v = np.asarray(verts, np.float64)
    x, y = v[:, 0], v[:, 1]
    x1, y1 = np.roll(x, -1), np.roll(y, -1)
    cr = x * y1 - x1 * y
    area = cr.sum() / 2.0
    cx = ((x + x1) * cr).sum() / (6.0 * area)
    cy = ((y + y1) * cr).sum() / (6.0 * area)
    ixx = ((y * y + y * y1 + y1 * y1) * cr).sum() / 12.0
    iyy = ((x * x + x * x1 + x1 * x1) * cr).sum() / 12.0
    mass = density * abs(area)
    inertia_origin = density * abs(ixx + iyy)
    return mass, (cx, cy), inertia_origin - mass * (cx * cx + cy * cy)


def masses():
    """(m[3], I[3]) of hull, leg, leg from gymnasium's shapes: LANDER_POLY / SCALE at density 5, boxes of half-extents
    (LEG_W, LEG_H) / SCALE at density 1."""
    mh, _, ih = polygon_mass_inertia([(x / SCALE, y / SCALE) for x, y in LANDER_POLY], 5.0)
    w2, h2 = 2 * LEG_W / SCALE, 2 * LEG_H / SCALE
    ml = 1.0 * w2 * h2
    il = ml * (w2 * w2 + h2 * h2) / 12.0
    return np.array([mh, ml, ml]), np.array([ih, il, il])


def momentum(world, m):
    """Total linear momentum [2, n] of the three bodies."""
    return (m[:, None, None] * world.v).sum(0)


def hull_position(world):
    """World position of the hull's body origin (gymnasium's lander.position): centre of mass minus the rotated local
    centroid (0, lcy)."""
    _, (cx, cy), _ = polygon_mass_inertia([(x / SCALE, y / SCALE) for x, y in LANDER_POLY], 5.0)
    s, c = np.sin(world.a[0]), np.cos(world.a[0])
    return world.c[0, 0] - (c * cx - s * cy), world.c[0, 1] - (s * cx + c * cy)


def engine_impulse(angle, action, u0, u1):
    """gymnasium LunarLander.step (discrete actions, no wind): the linear impulse applied to the hull, from the hull
    angle before the step and the step's two uniform draws u in [0, 1) (dispersion = (2u - 1) / SCALE)."""
    d0, d1 = (-1.0 + 2.0 * u0) / SCALE, (-1.0 + 2.0 * u1) / SCALE
    tip = (np.sin(angle), np.cos(angle))
    side = (-tip[1], tip[0])
    if action == 2:
        ox = tip[0] * (4 / SCALE + 2 * d0) + side[0] * d1
        oy = -tip[1] * (4 / SCALE + 2 * d0) - side[1] * d1
        return -ox * MAIN_ENGINE_POWER, -oy * MAIN_ENGINE_POWER
    if action in (1, 3):
        direction = action - 2
        ox = tip[0] * d0 + side[0] * (3 * d1 + direction * SIDE_ENGINE_AWAY / SCALE)
        oy = -tip[1] * d0 - side[1] * (3 * d1 + direction * SIDE_ENGINE_AWAY / SCALE)
        return -ox * SIDE_ENGINE_POWER, -oy * SIDE_ENGINE_POWER
    return 0.0 * angle, 0.0 * angle


def sleep_steps():
    """Number of consecutive quiet steps after which Box2D's island falls asleep: float32 accumulation of dt = 1/50
    against b2_timeToSleep = 0.5f."""
    h, t, k = np.float32(1.0) / np.float32(50.0), np.float32(0.0), 0
    while t < np.float32(0.5):
        t = np.float32(t + h)
        k += 1
    return k


def quiet(world):
    """b2Island::Solve's sleep test per env: every body under the linear (0.01 m/s) and angular (2 deg/s) tolerances."""
    lin = (world.v ** 2).sum(1) <= 0.01 ** 2
    ang = world.w ** 2 <= (2.0 / 180.0 * np.pi) ** 2
    return (lin & ang).all(0)


def heuristic(o):
    """gymnasium's lunar_lander.heuristic() for a batch of observations -> discrete actions."""
    angle_targ = np.clip(o[:, 0] * 0.5 + o[:, 2] * 1.0, -0.4, 0.4)
    hover_targ = 0.55 * np.abs(o[:, 0])
    angle_todo = (angle_targ - o[:, 4]) * 0.5 - o[:, 5] * 1.0
    hover_todo = (hover_targ - o[:, 1]) * 0.5 - o[:, 3] * 0.5
    legs = (o[:, 6] > 0) | (o[:, 7] > 0)
    angle_todo = np.where(legs, 0.0, angle_todo)
    hover_todo = np.where(legs, -o[:, 3] * 0.5, hover_todo)
    a = np.zeros(o.shape[0], np.int32)
    a = np.where((hover_todo > np.abs(angle_todo)) & (hover_todo > 0.05), 2, a)
    a = np.where((a == 0) & (angle_todo < -0.05), 3, a)
    a = np.where((a == 0) & (angle_todo > +0.05), 1, a)
    return a.astype(np.int32)


def run_scenarios(make, philox, gold, n=32, seed=11):
    """make(n, seed) -> (reset() -> obs, step(actions) -> dict(obs, rew, done, terminated), words() -> u32[144, n]).
    Runs every scenario and asserts against the closed-form expectations in `gold` (box2d_micro.npz)."""
    m = gold["mass"]
    Mg_dt = float(gold["Mg_dt"])
    # ---- B: free flight, momentum balance per step; E: joint limits once the legs are open
    reset, step, words = make(n, seed)
    reset()
    prev = World(words())
    for t in range(30):
        step(np.zeros(n, np.int32))
        cur = World(words())
        assert (cur.contacts == 0).all(), "free flight"
        dP = momentum(cur, m) - momentum(prev, m)
        assert np.abs(dP[0]).max() <= 2e-4, ("horizontal momentum is conserved", np.abs(dP[0]).max())
        assert np.abs(dP[1] + Mg_dt).max() <= 2e-4, ("vertical momentum changes by -M g dt", dP[1])
        prev = cur
    lo, hi, slop = gold["joint_lower"], gold["joint_upper"], float(gold["angular_slop"])
    for L in range(2):
        rel = prev.a[1 + L] - prev.a[0]
        assert ((rel >= lo[L] - 2 * slop) & (rel <= hi[L] + 2 * slop)).all(), (L, rel.min(), rel.max())
    # ---- C: engines.  Step index = steps taken in the episode so far (the reset's own step(0) uses index 0 too, with
    # the initial random force; the first agent step is index 0 again in this build's stream: use the draws the stepper
    # documents: Philox(seed, env, episode, RNG_ENV_STEP | ep_len))
    for action in (2, 1, 3):
        reset, step, words = make(n, seed + action)
        reset()
        prev = World(words())
        for t in range(12):
            step(np.full(n, action, np.int32))
            cur = World(words())
            dP = momentum(cur, m) - momentum(prev, m)
            for i in range(n):
                r = philox(seed + action, i, 0, 0, RNG_ENV_STEP | t)
                u0, u1 = (r[0] >> 8) * 2.0 ** -24, (r[1] >> 8) * 2.0 ** -24
                ix, iy = engine_impulse(prev.a[0, i], action, u0, u1)
                assert abs(dP[0, i] - ix) <= 3e-4 and abs(dP[1, i] - (iy - Mg_dt)) <= 3e-4, (action, t, i, dP[:, i], ix, iy - Mg_dt)
            prev = cur
    # ---- D: land with gymnasium's heuristic; sleep after exactly K quiet steps; normal impulses carry the weight
    K = int(gold["sleep_steps"])
    reset, step, words = make(n, seed + 7)
    o = reset()
    hist_q, hist_imp, hist_pos = [], [], []
    slept = np.zeros(n, bool)
    over = np.zeros(n, bool)
    checked = 0
    for t in range(1000):
        r = step(heuristic(o))
        w = World(words())
        q = quiet(w)
        # (after a done the words already hold the NEXT episode; `term_w` below is only read for envs still running)
        done = r["done"].astype(bool)
        landed_now = done & ~over & (r["rew"] == 100.0)
        hist_q.append(q), hist_imp.append(w.normal_impulse_sum), hist_pos.append(hull_position(w))
        for i in np.nonzero(landed_now)[0]:
            # the env ended on the K-th consecutive quiet step: steps t-K+1 .. t-1 were quiet (step t's own world has
            # been replaced by the reset), step t-K was not
            run = [hist_q[t - k][i] for k in range(1, K)]
            assert all(run), ("quiet run before sleep", i, t, run)
            assert t - K < 0 or not hist_q[t - K][i], ("sleep fired late", i, t)
            # resting on the flat pad for the last 10 steps: contact impulses carry the weight, no drift
            imp = np.array([hist_imp[t - k][i] for k in range(1, 11)])
            assert np.all(np.abs(imp - Mg_dt) <= 0.03 * Mg_dt), (i, imp, Mg_dt)
            px = np.array([hist_pos[t - k][0][i] for k in range(1, 11)])
            py = np.array([hist_pos[t - k][1][i] for k in range(1, 11)])
            assert np.ptp(px) <= 2e-3 and np.ptp(py) <= 2e-3, (i, np.ptp(px), np.ptp(py))
            checked += 1
        slept |= landed_now
        over |= done
        o = r["obs"]
        if over.all():
            break
    assert checked >= n // 2, f"only {checked} of {n} heuristic landings ended asleep"
    return checked
