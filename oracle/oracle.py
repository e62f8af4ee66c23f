"""ctypes/numpy front-end of the CPU oracle (oracle/libgymrl_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package gymrl_amd/ must never import it.
Every function takes and returns numpy arrays; see gymrl_oracle.c for the
reference file:line each one restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgymrl_oracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_expf.restype = C.c_float
        _lib.orc_expf.argtypes = [C.c_float]
        _lib.orc_logf.restype = C.c_float
        _lib.orc_logf.argtypes = [C.c_float]
        _lib.orc_tanhf.restype = C.c_float
        _lib.orc_tanhf.argtypes = [C.c_float]
        _lib.orc_env_create.restype = C.c_void_p
        _lib.orc_env_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------ math ----
def expf(x):
    L = lib()
    return np.array([L.orc_expf(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))


def logf(x):
    L = lib()
    return np.array([L.orc_logf(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))


def tanhf(x):
    L = lib()
    return np.array([L.orc_tanhf(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))


def linear_act(x, W, b, act=0):
    """One stage of gymrl_mlp_forward: act(x W^T + b) in the kernel's summation order."""
    x, W = _f32(x), _f32(W)
    n, in_dim = x.shape
    out_dim = W.shape[0]
    b = None if b is None else _f32(b)
    y = np.empty((n, out_dim), np.float32)
    lib().orc_linear_act(_p(x), _p(W), _p(b), n, in_dim, out_dim, int(act), _p(y))
    return y


def lunar_constants():
    out = np.zeros(6, np.float32)
    lib().orc_lunar_constants(_p(out))
    return out


def linear_fwd(x, W, b=None):
    """x W^T + b in gymrl_linear_fwd's accumulation order (no activation)."""
    x, W = _f32(x), _f32(W)
    B, K = x.shape
    N = W.shape[0]
    b = None if b is None else _f32(b)
    y = np.empty((B, N), np.float32)
    lib().orc_linear_fwd(_p(x), _p(W), _p(b), B, K, N, _p(y))
    return y


def linear_bwd_input(dy, W, H=None):
    """(dy W) * (1 - H^2) in gymrl_linear_bwd_input's accumulation order."""
    dy, W = _f32(dy), _f32(W)
    B, N = dy.shape
    K = W.shape[1]
    H = None if H is None else _f32(H)
    dx = np.empty((B, K), np.float32)
    lib().orc_linear_bwd_input(_p(dy), _p(W), _p(H), B, N, K, _p(dx))
    return dx


def linear_bwd_weight(dy, x, slices, rows_per_slice):
    """dy^T x in gymrl_linear_bwd_weight's slice / chain order."""
    dy, x = _f32(dy), _f32(x)
    B, N = dy.shape
    K = x.shape[1]
    dW = np.empty((N, K), np.float32)
    lib().orc_linear_bwd_weight(_p(dy), _p(x), C.c_int64(B), N, K, int(slices), C.c_int64(int(rows_per_slice)), _p(dW))
    return dW


def linear_bwd_bias(dy, slices, rows_per_slice):
    """Column sums of dy in gymrl_linear_bwd_weight's order (its `db` output)."""
    dy = _f32(dy)
    B, N = dy.shape
    db = np.empty(N, np.float32)
    lib().orc_linear_bwd_bias(_p(dy), C.c_int64(B), N, int(slices), C.c_int64(int(rows_per_slice)), _p(db))
    return db


def mlp_forward(x, stages):
    """stages: list of (W, b, act, src, dst); src -1 = x, dst -1 = network output.  Returns the list of
    outputs of the dst == -1 stages in order (include/gymrl.h gymrl_mlp_forward)."""
    bufs, outs = {-1: _f32(x)}, []
    for W, b, act, src, dst in stages:
        y = linear_act(bufs[src], W, b, act)
        if dst < 0:
            outs.append(y)
        else:
            bufs[dst] = y
    return outs


def sincosf(x):
    L = lib()
    xs = np.asarray(x, np.float32).ravel()
    s = np.empty_like(xs)
    c = np.empty_like(xs)
    ss, cc = C.c_float(), C.c_float()
    for i, v in enumerate(xs):
        L.orc_sincosf(C.c_float(float(v)), C.byref(ss), C.byref(cc))
        s[i], c[i] = ss.value, cc.value
    return s.reshape(np.shape(x)), c.reshape(np.shape(x))


def philox(key, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(C.c_uint64(key), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), out)
    return list(out)


# ------------------------------------------------------------------- GAE ----
def gae(rew, val, done, next_val, gamma, lam, want_moments=False):
    rew, val, done, next_val = _f32(rew), _f32(val), _u8(done), _f32(next_val)
    T, N = rew.shape
    adv, ret = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    mom = np.zeros(3, np.float64)
    lib().orc_gae(_p(rew), _p(val), _p(done), _p(next_val), C.c_int(T), C.c_int(N), C.c_double(gamma),
                  C.c_double(lam), _p(adv), _p(ret), _p(mom))
    return (adv, ret, mom) if want_moments else (adv, ret)


def gae_dw(rew, val, next_val, done, dw, gamma, lam):
    rew, val, next_val, done, dw = _f32(rew), _f32(val), _f32(next_val), _u8(done), _u8(dw)
    T, N = rew.shape
    adv, vt = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    mom = np.zeros(3, np.float64)
    lib().orc_gae_dw(_p(rew), _p(val), _p(next_val), _p(done), _p(dw), C.c_int(T), C.c_int(N),
                     C.c_double(gamma), C.c_double(lam), _p(adv), _p(vt), _p(mom))
    return adv, vt, mom


def gae_decoupled(rew, val, done, next_val, gamma, lam_actor, lam_critic):
    rew, val, done, next_val = _f32(rew), _f32(val), _u8(done), _f32(next_val)
    T, N = rew.shape
    adv, ret = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    lib().orc_gae_decoupled(_p(rew), _p(val), _p(done), _p(next_val), C.c_int(T), C.c_int(N),
                            C.c_double(gamma), C.c_double(lam_actor), C.c_double(lam_critic), _p(adv), _p(ret))
    return adv, ret


def moments(x):
    x = _f32(x).ravel()
    mom = np.zeros(3, np.float64)
    lib().orc_moments(_p(x), C.c_int64(x.size), _p(mom))
    return mom


def normalize(x, mom, ddof=0, eps=1e-8):
    x = _f32(x).copy()
    mom = np.ascontiguousarray(mom, np.float64)
    lib().orc_normalize(_p(x), C.c_int64(x.size), _p(mom), C.c_int(ddof), C.c_double(eps))
    return x


# ------------------------------------------------------------ categorical ---
def categorical_sample(logits, value=None, noise_exp=None, seed=0, counter=0, env_id0=0, deterministic=False):
    logits = _f32(logits)
    n, A = logits.shape
    act = np.empty(n, np.int32)
    logp, ent = np.empty(n, np.float32), np.empty(n, np.float32)
    vin = None if value is None else _f32(value)
    vout = None if value is None else np.empty(n, np.float32)
    q = None if noise_exp is None else _f32(noise_exp)
    lib().orc_categorical_sample(_p(logits), _p(vin), _p(q), C.c_uint64(seed), C.c_uint64(counter),
                                 C.c_int64(env_id0), C.c_int(n), C.c_int(A), C.c_int(int(deterministic)),
                                 _p(act), _p(logp), _p(ent), _p(vout))
    return act, logp, ent, vout


# --------------------------------------------------------------- PPO loss ---
class PPOCfg(C.Structure):
    _fields_ = [("clip_eps", C.c_float), ("dual_clip", C.c_float), ("value_coef", C.c_float),
                ("entropy_coef", C.c_float)]


class PPOFullCfg(C.Structure):
    _fields_ = [("clip_eps_min", C.c_float), ("clip_eps_max", C.c_float), ("dual_clip", C.c_float),
                ("erc_beta_low", C.c_float), ("erc_beta_high", C.c_float), ("entropy_coef", C.c_float)]


def ppo_loss_fwd_bwd(logits, value, act, logp_old, adv, ret, cfg, idx=None, adv_moments=None):
    logits, value = _f32(logits), _f32(value)
    B, A = logits.shape
    act, logp_old, adv, ret = _i32(act), _f32(logp_old), _f32(adv), _f32(ret)
    idx_ = None if idx is None else _i32(idx)
    mom = None if adv_moments is None else np.ascontiguousarray(adv_moments, np.float64)
    dl, dv = np.empty((B, A), np.float32), np.empty(B, np.float32)
    met = np.zeros(5, np.float64)
    c = PPOCfg(*cfg)
    lib().orc_ppo_loss_fwd_bwd(_p(logits), _p(value), _p(idx_), _p(act), _p(logp_old), _p(adv), _p(ret),
                               _p(mom), C.c_int(B), C.c_int(A), C.byref(c), _p(dl), _p(dv), _p(met))
    return dl, dv, met


def ppo_full_loss_fwd_bwd(logits, value, act, logp_old, ent_old, adv, ret, cfg, idx=None, corr_mul=None):
    logits, value = _f32(logits), _f32(value)
    B, A = logits.shape
    act, logp_old, ent_old, adv, ret = _i32(act), _f32(logp_old), _f32(ent_old), _f32(adv), _f32(ret)
    idx_ = None if idx is None else _i32(idx)
    dl, dv = np.empty((B, A), np.float32), np.empty(B, np.float32)
    met = np.zeros(9, np.float64)
    c = PPOFullCfg(*cfg)
    lib().orc_ppo_full_loss_fwd_bwd(_p(logits), _p(value), _p(idx_), _p(act), _p(logp_old), _p(ent_old),
                                    _p(adv), _p(ret), C.c_int(B), C.c_int(A), C.byref(c),
                                    _p(None if corr_mul is None else _f32(corr_mul)), _p(dl), _p(dv), _p(met))
    return dl, dv, met


def ppo_rnn_loss_fwd_bwd(logits, value, act, logp_old, ent_old, val_old, adv, ret, cfg, idx=None, corr_mul=None):
    logits, value = _f32(logits), _f32(value)
    B, A = logits.shape
    act, logp_old, ent_old, val_old, adv, ret = _i32(act), _f32(logp_old), _f32(ent_old), _f32(val_old), _f32(adv), _f32(ret)
    idx_ = None if idx is None else _i32(idx)
    dl, dv = np.empty((B, A), np.float32), np.empty(B, np.float32)
    met = np.zeros(10, np.float64)
    c = PPOFullCfg(*cfg)
    lib().orc_ppo_rnn_loss_fwd_bwd(_p(logits), _p(value), _p(idx_), _p(act), _p(logp_old), _p(ent_old), _p(val_old),
                                   _p(adv), _p(ret), C.c_int(B), C.c_int(A), C.byref(c),
                                   _p(None if corr_mul is None else _f32(corr_mul)), _p(dl), _p(dv), _p(met))
    return dl, dv, met


def gru_cell_fwd(gi, gh, h):
    gi, gh, h = _f32(gi), _f32(gh), _f32(h)
    B, H = h.shape
    out = np.empty_like(h)
    lib().orc_gru_cell_fwd(_p(gi), _p(gh), _p(h), C.c_int(B), C.c_int(H), _p(out))
    return out


def gru_cell_bwd(gi, gh, h, dh_out):
    gi, gh, h, dh_out = _f32(gi), _f32(gh), _f32(h), _f32(dh_out)
    B, H = h.shape
    dgi, dgh, dh = np.empty_like(gi), np.empty_like(gh), np.empty_like(h)
    lib().orc_gru_cell_bwd(_p(gi), _p(gh), _p(h), _p(dh_out), C.c_int(B), C.c_int(H), _p(dgi), _p(dgh), _p(dh))
    return dgi, dgh, dh


def gru_forward(x, h0, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU(batch_first=True, one layer) over x [B, L, D] from h0 [B, H] -> (out [B, L, H], h_L)."""
    x, h = _f32(x), _f32(h0)
    outs = []
    for l in range(x.shape[1]):
        gi = (x[:, l].astype(np.float64) @ _f32(w_ih).T.astype(np.float64) + b_ih).astype(np.float32)
        gh = (h.astype(np.float64) @ _f32(w_hh).T.astype(np.float64) + b_hh).astype(np.float32)
        h = gru_cell_fwd(gi, gh, h)
        outs.append(h)
    return np.stack(outs, 1), h


def rnd_reward(predict, target, rew=None):
    predict, target = _f32(predict), _f32(target)
    B, E = predict.shape
    rnd = np.empty(B, np.float32)
    rew_ = None if rew is None else _f32(rew).copy()
    lib().orc_rnd_reward(_p(predict), _p(target), C.c_int(B), C.c_int(E), _p(rew_), _p(rnd))
    return rnd, rew_


def permutation(seed, counter, M):
    out = np.empty(M, np.int32)
    lib().orc_permutation(C.c_uint64(seed), C.c_uint64(counter), C.c_int64(M), _p(out))
    return out


def pack_rollout(obs, act, logp, adv, ret):
    """ppo_lunarlander.py:238-250 staging as one 64-B record per transition (numpy restatement)."""
    obs = _f32(obs)
    M, D = obs.shape
    rec = np.zeros((M, 16), np.float32)
    rec[:, :D] = obs
    rec[:, 12] = _i32(act).view(np.float32)
    rec[:, 13], rec[:, 14], rec[:, 15] = _f32(logp), _f32(adv), _f32(ret)
    return rec


def gather_minibatch(packed, idx, obs_dim):
    """ppo_lunarlander.py:264-272: rows of the shuffled index slice."""
    r = packed[np.asarray(idx, np.int64)]
    return (np.ascontiguousarray(r[:, :obs_dim]), np.ascontiguousarray(r[:, 12]).view(np.int32), r[:, 13].copy(),
            r[:, 14].copy(), r[:, 15].copy())


# -------------------------------------------------------------- optimiser ---
def sqnorm(g, grad_scale=1.0):
    g = _f32(g).ravel()
    out = np.zeros(1, np.float64)
    lib().orc_sqnorm(_p(g), C.c_int64(g.size), C.c_float(grad_scale), _p(out))
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, max_grad_norm=0.0, clamp_abs=0.0,
              zero_grad=False):
    """In place on copies; returns (p, g, m, v)."""
    p, g, m, v = [_f32(a).ravel().copy() for a in (p, g, m, v)]
    sq = sqnorm(g, grad_scale) if max_grad_norm > 0 else None
    lib().orc_adam_step(_p(p), _p(g), _p(m), _p(v), C.c_int64(p.size), C.c_double(lr), C.c_double(beta1),
                        C.c_double(beta2), C.c_double(eps), C.c_int64(step), C.c_float(grad_scale),
                        C.c_float(max_grad_norm), _p(sq), C.c_float(clamp_abs), C.c_int(int(zero_grad)))
    return p, g, m, v


def soft_update(target, source, tau):
    t = _f32(target).ravel().copy()
    s = _f32(source).ravel()
    lib().orc_soft_update(_p(t), _p(s), C.c_int64(t.size), C.c_double(tau))
    return t


# -------------------------------------------------------------------- env ---
CARTPOLE, PENDULUM, LUNARLANDER = 0, 1, 2
_OBS = {0: 4, 1: 3, 2: 8}


class Env:
    """Vector of oracle envs with the same auto-reset contract as gymrl_env_step."""

    def __init__(self, kind, n, seed=0, env_id0=0):
        self.kind, self.n, self.D = kind, n, _OBS[kind]
        self._h = C.c_void_p(lib().orc_env_create(kind, n, C.c_uint64(seed), C.c_int64(env_id0)))

    def __del__(self):
        try:
            lib().orc_env_destroy(self._h)
        except Exception:
            pass

    def reset(self):
        obs = np.zeros((self.n, self.D), np.float32)
        lib().orc_env_reset(self._h, _p(obs))
        return obs

    def abandon(self, cap, obs):
        """gymrl_env_abandon: returns (obs with the restarted envs' rows replaced, flag u8[n], ep_ret, ep_len)."""
        obs = np.ascontiguousarray(obs, np.float32).copy()
        flag, ep_ret, ep_len = np.zeros(self.n, np.uint8), np.zeros(self.n, np.float32), np.zeros(self.n, np.int32)
        lib().orc_env_abandon(self._h, int(cap), _p(obs), _p(flag), _p(ep_ret), _p(ep_len))
        return obs, flag, ep_ret, ep_len

    def set_classic_state(self, states, ep_len=0):
        """Test hook: env i := states[i] (f64; CartPole x, xdot, th, thdot / Pendulum th, thdot), `ep_len` steps taken."""
        states = np.ascontiguousarray(states, np.float64)
        lens = np.broadcast_to(np.asarray(ep_len, np.int64), (self.n,))
        for i in range(self.n):
            row = np.ascontiguousarray(states[i])
            lib().orc_env_set_classic(self._h, i, _p(row), int(lens[i]))

    def lunar_words(self):
        """u32[144, n]: every LunarLander world in the word order of the HIP state buffer (tests/box2d_micro.py)."""
        out = np.zeros((self.n, 144), np.uint32)
        row = np.zeros(144, np.uint32)
        for i in range(self.n):
            lib().orc_env_lunar_words(self._h, i, _p(row))
            out[i] = row
        return np.ascontiguousarray(out.T)

    def step(self, action):
        action = _i32(action) if self.kind != PENDULUM else _f32(action)
        n, D = self.n, self.D
        obs, tobs = np.zeros((n, D), np.float32), np.zeros((n, D), np.float32)
        rew = np.zeros(n, np.float32)
        term, trunc, done = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        ep_ret, ep_len = np.zeros(n, np.float32), np.zeros(n, np.int32)
        stats = np.zeros(3, np.float64)
        lib().orc_env_step(self._h, _p(action), _p(obs), _p(tobs), _p(rew), _p(term), _p(trunc), _p(done),
                           _p(ep_ret), _p(ep_len), _p(stats))
        return dict(obs=obs, term_obs=tobs, rew=rew, terminated=term, truncated=trunc, done=done,
                    ep_ret=ep_ret, ep_len=ep_len, ep_stats=stats)


# =============================================================== off-policy ===
def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class SumTree:
    """rainbow_dqn_cartpole.py:116-152 on a float64 numpy array (C loops)."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.tree = np.zeros(2 * self.capacity - 1, np.float64)

    def update(self, data_index, priority):
        lib().orc_tree_update(_p(self.tree), C.c_int64(self.capacity), C.c_int64(int(data_index)), C.c_double(priority))

    def update_many(self, idx=None, prio=None, idx_start=0, idx_is_tree=False, prio_scalar=0.0, B=None):
        idx_ = None if idx is None else _i32(idx)
        pr = None if prio is None else _f64(prio)
        B = (idx_.size if idx_ is not None else (pr.size if pr is not None else B))
        lib().orc_tree_update_many(_p(self.tree), C.c_int64(self.capacity), _p(idx_), C.c_int64(idx_start),
                                   C.c_int(int(idx_is_tree)), _p(pr), C.c_double(prio_scalar), C.c_int(B))

    def max_leaf(self):
        lib().orc_tree_max_leaf.restype = C.c_double
        return lib().orc_tree_max_leaf(_p(self.tree), C.c_int64(self.capacity))

    def sample(self, B, size, beta, u=None, seed=0, counter=0, variant_b=False):
        u_ = None if u is None else _f64(u)
        idx, prio, w = np.empty(B, np.int32), np.empty(B, np.float64), np.empty(B, np.float32)
        lib().orc_per_sample(_p(self.tree), C.c_int64(self.capacity), _p(u_), C.c_uint64(seed), C.c_uint64(counter),
                             C.c_int(B), C.c_int64(size), C.c_double(beta), C.c_int(int(variant_b)), _p(idx),
                             _p(prio), _p(w))
        return idx, prio, w


def per_priorities(td, alpha, eps, clip=0.0):
    td = _f32(td)
    out = np.empty(td.size, np.float64)
    lib().orc_per_priorities(_p(td), C.c_int(td.size), C.c_double(alpha), C.c_double(eps), C.c_double(clip), _p(out))
    return out


def uniform_indices(seed, counter, size, B):
    idx = np.empty(B, np.int32)
    lib().orc_uniform_indices(C.c_uint64(seed), C.c_uint64(counter), C.c_int64(size), C.c_int(B), _p(idx))
    return idx


class ReplayRing:
    """numpy restatement of the SoA ring (dqn_cartpole.py:68-88 / sac_pendulum.py:128-148)."""

    def __init__(self, cap, D, AW=1):
        self.cap, self.D, self.AW = cap, D, AW
        self.state, self.next_state = np.zeros((cap, D), np.float32), np.zeros((cap, D), np.float32)
        self.action = np.zeros((cap, AW), np.uint32)
        self.reward, self.flag = np.zeros(cap, np.float32), np.zeros(cap, np.uint8)
        self.cursor, self.size = 0, 0

    def append(self, s, a, r, s2, f):
        n = len(r)
        rows = (self.cursor + np.arange(n)) % self.cap
        self.state[rows], self.next_state[rows] = s, s2
        self.action[rows] = np.ascontiguousarray(a).view(np.uint32).reshape(n, self.AW)
        self.reward[rows], self.flag[rows] = r, f
        self.cursor = (self.cursor + n) % self.cap
        self.size = min(self.size + n, self.cap)

    def gather(self, idx):
        idx = np.asarray(idx, np.int64)
        return (self.state[idx], self.action[idx], self.reward[idx], self.next_state[idx],
                self.flag[idx].astype(np.float32))


class NStepWindows:
    """rainbow_dqn_cartpole.py:179-218 for N env windows feeding a ReplayRing."""

    def __init__(self, n_steps, N, D, gamma):
        self.n, self.N, self.D, self.gamma, self.pushes = n_steps, N, D, gamma, 0
        self.w_state, self.w_next = np.zeros((n_steps, N, D), np.float32), np.zeros((n_steps, N, D), np.float32)
        self.w_action, self.w_reward = np.zeros((n_steps, N), np.int32), np.zeros((n_steps, N), np.float32)
        self.w_terminal, self.w_done = np.zeros((n_steps, N), np.uint8), np.zeros((n_steps, N), np.uint8)

    def push(self, ring, obs, action, reward, next_obs, terminal, done):
        obs, next_obs, reward = _f32(obs), _f32(next_obs), _f32(reward)
        action, terminal, done = _i32(action), _u8(terminal), _u8(done)
        emit = lib().orc_nstep_push(_p(self.w_state), _p(self.w_action), _p(self.w_reward), _p(self.w_next),
                                    _p(self.w_terminal), _p(self.w_done), C.c_int(self.n), C.c_int64(self.pushes),
                                    C.c_int(self.N), C.c_int(self.D), C.c_double(self.gamma), _p(obs), _p(action),
                                    _p(reward), _p(next_obs), _p(terminal), _p(done), _p(ring.state),
                                    _p(ring.action), _p(ring.reward), _p(ring.next_state), _p(ring.flag),
                                    C.c_int64(ring.cap), C.c_int64(ring.cursor))
        self.pushes += 1
        if emit:
            ring.cursor = (ring.cursor + self.N) % ring.cap
            ring.size = min(ring.size + self.N, ring.cap)
        return bool(emit)


def noisy_noise(nin, nout, eps_in=None, eps_out=None, seed=0, counter=0):
    ei = None if eps_in is None else _f32(eps_in)
    eo = None if eps_out is None else _f32(eps_out)
    w, b = np.empty((nout, nin), np.float32), np.empty(nout, np.float32)
    lib().orc_noisy_noise(_p(ei), _p(eo), C.c_uint64(seed), C.c_uint64(counter), C.c_int(nin), C.c_int(nout), _p(w), _p(b))
    return w, b


def noisy_action(mu, std, bound, eps=None, mode=0, noise_clip=0.0, seed=0, counter=0):
    mu = _f32(mu)
    e = None if eps is None else np.ascontiguousarray(eps, dtype=np.float64)
    out = np.empty_like(mu)
    lib().orc_noisy_action(_p(mu), _p(e), C.c_uint64(seed), C.c_uint64(counter), C.c_int64(mu.size), C.c_int(mode),
                           C.c_double(std), C.c_double(noise_clip), C.c_double(bound), _p(out))
    return out


def mse_loss(q, y):
    q, y = _f32(q), _f32(y)
    dq, s = np.empty_like(q), np.zeros(1, np.float64)
    lib().orc_mse_loss(_p(q), _p(y), C.c_int(q.size), _p(dq), _p(s))
    return dq, s[0]


def neg_mean_loss(q):
    q = _f32(q)
    dq, s = np.empty_like(q), np.zeros(1, np.float64)
    lib().orc_neg_mean_loss(_p(q), C.c_int(q.size), _p(dq), _p(s))
    return dq, s[0]


def dsac_target(probs_n, q1n, q2n, rew, done, log_alpha, gamma):
    probs_n, q1n, q2n, rew, done = (_f32(a) for a in (probs_n, q1n, q2n, rew, done))
    B, A = probs_n.shape
    y, la = np.empty(B, np.float32), np.array([log_alpha], np.float32)
    lib().orc_dsac_target(_p(probs_n), _p(q1n), _p(q2n), _p(rew), _p(done), _p(la), B, A, C.c_float(gamma), _p(y))
    return y


def dsac_critic_loss(q1, q2, act, y):
    q1, q2, y, act = _f32(q1), _f32(q2), _f32(y), _i32(act)
    B, A = q1.shape
    d1, d2, s = np.empty_like(q1), np.empty_like(q2), np.zeros(2, np.float64)
    lib().orc_dsac_critic_loss(_p(q1), _p(q2), _p(act), _p(y), B, A, _p(d1), _p(d2), _p(s))
    return d1, d2, s


def dsac_actor_loss(probs, q1, q2, log_alpha):
    probs, q1, q2 = _f32(probs), _f32(q1), _f32(q2)
    B, A = probs.shape
    dp, s, la = np.empty_like(probs), np.zeros(2, np.float64), np.array([log_alpha], np.float32)
    lib().orc_dsac_actor_loss(_p(probs), _p(q1), _p(q2), _p(la), B, A, _p(dp), _p(s))
    return dp, s


def dsac_alpha_step(log_alpha, m, v, sums, B, target_entropy, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    la, mm, vv = np.array([log_alpha], np.float32), np.array([m], np.float32), np.array([v], np.float32)
    loss = np.zeros(1, np.float64)
    lib().orc_dsac_alpha_step(_p(la), _p(mm), _p(vv), _p(np.ascontiguousarray(sums, np.float64)), B, C.c_float(target_entropy),
                              C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_int64(step), _p(loss))
    return float(la[0]), float(mm[0]), float(vv[0]), float(loss[0])


def epsilon_greedy(q, epsilon, u=None, seed=0, counter=0, env_id0=0):
    q = _f32(q)
    n, A = q.shape
    u_ = None if u is None else _f32(u)
    act = np.empty(n, np.int32)
    lib().orc_epsilon_greedy(_p(q), _p(u_), C.c_uint64(seed), C.c_uint64(counter), C.c_int64(env_id0), C.c_int(n),
                             C.c_int(A), C.c_float(epsilon), _p(act))
    return act


def dqn_td_loss(q, q_next_target, act, rew, flag, gamma_n, q_next_online=None, w=None):
    q, qt = _f32(q), _f32(q_next_target)
    B, A = q.shape
    qo = None if q_next_online is None else _f32(q_next_online)
    w_ = None if w is None else _f32(w)
    td, dq, loss = np.empty(B, np.float32), np.empty((B, A), np.float32), np.zeros(1, np.float64)
    lib().orc_dqn_td_loss(_p(q), _p(qo), _p(qt), _p(_i32(act)), _p(_f32(rew)), _p(_f32(flag)), _p(w_), C.c_int(B),
                          C.c_int(A), C.c_double(gamma_n), _p(td), _p(dq), _p(loss))
    return td, dq, loss


def sac_sample_fwd(mean, log_std, eps, bound):
    mean, log_std, eps = _f32(mean), _f32(log_std), _f32(eps)
    B, A = mean.shape
    act, logp = np.empty((B, A), np.float32), np.empty(B, np.float32)
    lib().orc_sac_sample_fwd(_p(mean), _p(log_std), _p(eps), C.c_int(B), C.c_int(A), C.c_float(bound), _p(act), _p(logp))
    return act, logp


def sac_sample_bwd(mean, log_std, eps, d_action, d_logp, bound):
    mean, log_std, eps = _f32(mean), _f32(log_std), _f32(eps)
    B, A = mean.shape
    da = None if d_action is None else _f32(d_action)
    dl = None if d_logp is None else _f32(d_logp)
    dm, ds = np.empty((B, A), np.float32), np.empty((B, A), np.float32)
    lib().orc_sac_sample_bwd(_p(mean), _p(log_std), _p(eps), _p(da), _p(dl), C.c_int(B), C.c_int(A), C.c_float(bound),
                             _p(dm), _p(ds))
    return dm, ds


def sac_target(rew, done, q1n, q2n, logp_n, log_alpha, gamma):
    B = len(rew)
    y = np.empty(B, np.float32)
    la = np.array([log_alpha], np.float64)
    lib().orc_sac_target(_p(_f32(rew)), _p(_f32(done)), _p(_f32(q1n)), _p(_f32(q2n)), _p(_f32(logp_n)), _p(la),
                         C.c_int(B), C.c_double(gamma), _p(y))
    return y


def sac_critic_loss(q1, q2, y):
    B = len(y)
    d1, d2, sums = np.empty(B, np.float32), np.empty(B, np.float32), np.zeros(4, np.float64)
    lib().orc_sac_critic_loss(_p(_f32(q1)), _p(_f32(q2)), _p(_f32(y)), C.c_int(B), _p(d1), _p(d2), _p(sums))
    return d1, d2, sums


def sac_actor_loss(logp, q1, q2, log_alpha, target_entropy):
    B = len(logp)
    dl, d1, d2, sums = (np.empty(B, np.float32) for _ in range(3)), None, None, None
    dl, d1, d2 = np.empty(B, np.float32), np.empty(B, np.float32), np.empty(B, np.float32)
    sums = np.zeros(4, np.float64)
    la = np.array([log_alpha], np.float64)
    lib().orc_sac_actor_loss(_p(_f32(logp)), _p(_f32(q1)), _p(_f32(q2)), _p(la), C.c_int(B), C.c_double(target_entropy),
                             _p(dl), _p(d1), _p(d2), _p(sums))
    return dl, d1, d2, sums


def sac_alpha_step(log_alpha, m, v, sums, B, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=1):
    la, m_, v_ = np.array([log_alpha], np.float64), np.array([m], np.float64), np.array([v], np.float64)
    loss = np.zeros(1, np.float64)
    lib().orc_sac_alpha_step(_p(la), _p(m_), _p(v_), _p(_f64(sums)), C.c_int(B), C.c_double(lr), C.c_double(beta1),
                             C.c_double(beta2), C.c_double(eps), C.c_int64(step), _p(loss))
    return la[0], m_[0], v_[0], loss[0]


def running_norm_stats(D):
    return np.zeros(2 + 3 * D, np.float64)


def running_norm(x, stats, update=True):
    x = _f32(x)
    N, D = x.shape
    y = np.empty((N, D), np.float32)
    lib().orc_running_norm(_p(x), C.c_int(N), C.c_int(D), _p(stats), C.c_int(int(update)), _p(y))
    return y


def reward_scaling(r, done, gamma, R, stats):
    r = _f32(r)
    y = np.empty(r.size, np.float32)
    d = None if done is None else _u8(done)
    lib().orc_reward_scaling(_p(r), _p(d), C.c_int(r.size), C.c_double(gamma), _p(R), _p(stats), _p(y))
    return y
