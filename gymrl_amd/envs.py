"""Device-resident batched environments (CartPole-v1 / Pendulum-v1 / LunarLander-v3).

Stands where `gym.make(name)` stands in the reference trainers
(ppo_lunarlander.py:160, dqn_cartpole.py:94, rainbow_dqn_cartpole.py:270,
sac_pendulum.py:154, utils/runner.py:53).  N env instances live in one SoA state
buffer in HBM and are stepped by one HIP kernel launch (one env per lane) with
auto-reset; observations, rewards and flags never leave the device.
"""
import types

import torch

from . import ops


class _Space:
    def __init__(self, shape=None, n=None, high=None):
        self.shape, self.n, self.high = shape, n, high


class VecEnv:
    """N lane-parallel env instances.  Mirrors the parts of the gymnasium surface the
    reference touches: observation_space.shape, action_space.n|shape|high,
    spec.max_episode_steps, reset(seed), step(action)."""

    def __init__(self, env_name, num_envs, device="cuda:0", seed=0, env_id0=0, prefetch_resets=True):
        if env_name not in ops.ENV_KINDS:
            raise ValueError(f"unsupported env {env_name!r}; have {sorted(ops.ENV_KINDS)}")
        self.name, self.kind, self.n = env_name, ops.ENV_KINDS[env_name], int(num_envs)
        self.device = torch.device(device)
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.obs_dim, self.act_dim, self.discrete, self.max_steps = ops.env_dims(self.kind)
        self.observation_space = _Space(shape=(self.obs_dim,))
        self.action_space = (_Space(n=self.act_dim) if self.discrete
                             else _Space(shape=(self.act_dim,), high=[2.0] * self.act_dim))
        self.spec = types.SimpleNamespace(max_episode_steps=self.max_steps, id=env_name)
        self.state = ops.env_state(self.kind, self.n, self.device)
        d, n = self.device, self.n
        self.terminated = torch.zeros(n, dtype=torch.uint8, device=d)
        self.truncated = torch.zeros(n, dtype=torch.uint8, device=d)
        self.ep_stats = torch.zeros(3, dtype=torch.float64, device=d)  # (#episodes, sum return, sum len)
        # expensive resets (LunarLander) are prepared on a side stream, off the step's critical path
        self._side = torch.cuda.Stream(device=d) if (self.kind == ops.LUNARLANDER and prefetch_resets) else None

    def reset(self, obs_out=None, seed=None):
        if seed is not None:
            self.seed = int(seed)
        obs_out = torch.empty(self.n, self.obs_dim, device=self.device) if obs_out is None else obs_out
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)      # no refill may still be writing spares
        ops.env_reset(self.kind, self.state, self.n, self.seed, self.env_id0, obs_out)
        self._refill()
        return obs_out

    def _refill(self):
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())      # after the step/reset just queued
            ops.env_refill(self.kind, self.state, self.n, self.seed, self.env_id0, stream=self._side)

    def step(self, action, obs_out, rew_out, done_out=None, term_obs_out=None, ep_ret_out=None,
             ep_len_out=None, terminated_out=None, truncated_out=None):
        ops.env_step(self.kind, self.state, self.n, self.seed, self.env_id0, action, obs_out, rew_out,
                     self.terminated if terminated_out is None else terminated_out,
                     self.truncated if truncated_out is None else truncated_out,
                     term_obs_out=term_obs_out, done_out=done_out, ep_ret_out=ep_ret_out,
                     ep_len_out=ep_len_out, ep_stats=self.ep_stats)
        self._refill()

    def abandon(self, cap, obs_inout, flag_inout=None, ep_ret_out=None, ep_len_out=None):
        """A trainer's own step cap below the env's TimeLimit (`for step in range(cfg.max_steps)`): envs whose episode
        reached `cap` steps start their next one without a done flag; `flag_inout` (the tracker's done row) gets them too."""
        ops.env_abandon(self.kind, self.state, self.n, self.seed, self.env_id0, int(cap), obs_inout, flag_inout, ep_ret_out,
                        ep_len_out, self.ep_stats)

    @property
    def gym(self):
        """The gymnasium single-env API over this env (num_envs == 1 only): see GymView."""
        if getattr(self, "_gym", None) is None:
            self._gym = GymView(self)
        return self._gym

    def close(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GymView:
    """gymnasium's single-env API over a VecEnv of ONE env, for callers written against the reference's loop
    (`state, _ = env.reset()` ... `next_state, reward, terminated, truncated, _ = env.step(action)` ... `env.reset()` after
    a done: dqn_cartpole.py:176-189, sac_pendulum.py:275-292, ppo_lunarlander.py:200-223): numpy observations, python
    scalars.  The stepper auto-resets inside the kernel, gymnasium does not: step() therefore returns the TERMINAL
    observation of a finished episode and the reset() that the caller issues next hands out the first observation of the
    episode the kernel has already started (a reset(seed=...) with an explicit seed, or one in the middle of an episode,
    is a real reset).  `trainer.env.gym` gives the view; every call is one launch + one small D2H copy."""

    def __init__(self, env):
        if env.n != 1:
            raise ValueError("GymView is the single-env surface: num_envs must be 1")
        self.env = env
        self.observation_space, self.action_space, self.spec = env.observation_space, env.action_space, env.spec
        d = env.device
        self._obs = torch.empty(1, env.obs_dim, device=d)
        self._term = torch.empty(1, env.obs_dim, device=d)
        self._rew = torch.empty(1, device=d)
        self._flags = torch.zeros(2, 1, dtype=torch.uint8, device=d)
        self._next_ready = False
        self._resets = 0

    def reset(self, seed=None, options=None):
        if seed is not None or not self._next_ready:
            if seed is None and self._resets:                        # a fresh episode, not a replay of the first one
                seed = (self.env.seed + 0x9E3779B1 * self._resets) & 0x7FFFFFFFFFFFFFFF
            self.env.reset(self._obs, seed=seed)
        self._next_ready = False
        self._resets += 1
        return self._obs[0].cpu().numpy().copy(), {}

    def step(self, action):
        import numpy as np
        d = self.env.device
        if self.env.discrete:
            act = torch.tensor([int(action)], dtype=torch.int32, device=d)
        else:
            act = torch.from_numpy(np.asarray(action, np.float32).reshape(1, -1)).to(d)
        self.env.step(act, self._obs, self._rew, term_obs_out=self._term, terminated_out=self._flags[0],
                      truncated_out=self._flags[1])
        terminated, truncated = (bool(v) for v in self._flags.view(-1).tolist())
        done = terminated or truncated
        self._next_ready = done
        obs = (self._term if done else self._obs)[0].cpu().numpy().copy()
        return obs, float(self._rew.item()), terminated, truncated, {}

    def close(self):
        self.env.close()


def make(env_name, num_envs=1, **kw):
    return VecEnv(env_name, num_envs, **kw)


class EpisodeTracker:
    """Finished-episode returns without a host sync per step: (ep_ret, done) rows are kept
    on the device and drained every `flush_every` vector steps into a python deque, in
    time order (the reference appends episode_reward at every done)."""

    def __init__(self, num_envs, device, flush_every=16):
        self.K = flush_every
        self.ret = torch.zeros(flush_every, num_envs, device=device)
        self.done = torch.zeros(flush_every, num_envs, dtype=torch.uint8, device=device)
        self.len = torch.zeros(flush_every, num_envs, dtype=torch.int32, device=device)
        self.lengths = []              # finished-episode lengths, parallel to the sink (filled when len_slot() is used)
        self._want_len = False
        self.k = 0
        self.episodes = 0

    def len_slot(self):
        """-> ep_len_out[N] view for the current vector step (optional: the legacy runner logs episode lengths)."""
        self._want_len = True
        return self.len[self.k]

    def slot(self):
        """-> (ep_ret_out[N], done_out[N]) views for the current vector step."""
        return self.ret[self.k], self.done[self.k]

    def advance(self, sink):
        self.k += 1
        if self.k == self.K:
            self.flush(sink)

    def drain_async(self):
        """Enqueue the device -> pinned-host copy of the filled rows behind whatever produces them and return a token for
        collect(); the rows may be overwritten by later steps at once (stream order).  No host sync, no device-side
        compaction (the boolean-mask indexing this replaces was three launches and a sync per drain)."""
        if not self.k:
            return None
        if getattr(self, "_pin", None) is None:
            K, N = self.ret.shape
            self._pin = [(torch.zeros(K, N, pin_memory=True), torch.zeros(K, N, dtype=torch.uint8, pin_memory=True),
                          torch.zeros(K, N, dtype=torch.int32, pin_memory=True)) for _ in range(2)]
            self._pp = 0
        self._pp ^= 1
        k, (pr, pd, pl) = self.k, self._pin[self._pp]
        pr[:k].copy_(self.ret[:k], non_blocking=True)
        pd[:k].copy_(self.done[:k], non_blocking=True)
        if self._want_len:
            pl[:k].copy_(self.len[:k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.k = 0
        return (self._pp, k, ev)

    def collect(self, token, sink):
        """Wait for a drain_async() copy and append its finished episodes to `sink`, in time order (row-major: step, then env)."""
        if token is None:
            return
        i, k, ev = token
        ev.synchronize()
        pr, pd, pl = self._pin[i]
        mask = pd[:k].numpy().astype(bool)
        fin = pr[:k].numpy()[mask].tolist()
        for r in fin:
            sink.append(r)
        if self._want_len:
            self.lengths.extend(pl[:k].numpy()[mask].tolist())
        self.episodes += len(fin)

    def flush(self, sink):
        self.collect(self.drain_async(), sink)
