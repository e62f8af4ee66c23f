"""Build-owned scripted deterministic environment.

Observations, rewards and episode ends are closed-form integer functions of
(episode#, t, action) -- no physics, no RNG, every value exactly representable in
float32 -- so the reference's trainers (driven through a stub `gymnasium.make` in
tests/golden/make_golden.py) and this repo's trainers (driven through
ScriptedVecEnv on the GPU) see bit-identical transitions.  What the traces pin is
therefore the trainers' *control flow*: per-rollout forced reset, reset observation as
the next policy input, done = terminated | truncated, bootstrap from the post-rollout
state, shuffled contiguous minibatches, LR anneal, episode-return bookkeeping.
"""
import types

import numpy as np


class ScriptedEnv:
    """gymnasium-style single env: reset(seed) -> (obs, info); step(a) -> 5-tuple."""

    def __init__(self, obs_dim=8, n_actions=4, continuous=False):
        self.obs_dim, self.n_actions, self.continuous = obs_dim, n_actions, continuous
        self.observation_space = types.SimpleNamespace(shape=(obs_dim,))
        if continuous:
            self.action_space = types.SimpleNamespace(shape=(1,), high=np.array([2.0], np.float32))
        else:
            self.action_space = types.SimpleNamespace(n=n_actions, sample=lambda: 0)
        self.spec = types.SimpleNamespace(max_episode_steps=500)
        self.episode, self.t, self.last_a = -1, 0, 0

    def _obs(self):
        k = np.arange(self.obs_dim, dtype=np.int64)
        h = (self.episode * 131 + self.t * 17 + k * 29 + self.last_a * 7) % 97
        return (h.astype(np.float32) / np.float32(97.0) - np.float32(0.5)).astype(np.float32)

    def episode_length(self):
        return 9 + (self.episode * 5) % 13

    def reset(self, seed=None, options=None):
        self.episode += 1
        self.t, self.last_a = 0, 0
        return self._obs(), {}

    def step(self, action):
        if self.continuous:      # quantise a bounded real action to 0..8 (boundaries every 0.5)
            a = int(min(8, max(0, np.floor((float(np.asarray(action).reshape(-1)[0]) + 2.0) * 2.0))))
        else:
            a = int(action)
        self.t += 1
        self.last_a = a
        reward = float(((self.episode * 7 + self.t * 3 + a * 11) % 23) - 11) / 4.0
        over = self.t >= self.episode_length()
        terminated = bool(over and self.episode % 3 != 2)
        truncated = bool(over and self.episode % 3 == 2)
        return self._obs(), reward, terminated, truncated, {}

    def close(self):
        pass


class ScriptedVecEnv:
    """The same script behind gymrl_amd.envs.VecEnv's device interface (N independent copies;
    copy i starts `i` resets ahead so lanes differ).  Host-computed: a test fixture, not a product path."""

    def __init__(self, num_envs, device, obs_dim=8, n_actions=4, episode0=None, continuous=False):
        import torch
        self.torch = torch
        self.n, self.device = num_envs, device
        self.envs = [ScriptedEnv(obs_dim, n_actions, continuous) for _ in range(num_envs)]
        for i, e in enumerate(self.envs):
            # training: lanes far apart; evaluation (episode0 given): copy i plays episode episode0 + i,
            # the i-th of the reference's sequential evaluation episodes
            e.episode += i * 1000 if episode0 is None else episode0 + i
        self.obs_dim, self.act_dim, self.discrete, self.max_steps = obs_dim, n_actions, True, 500
        self.spec = types.SimpleNamespace(max_episode_steps=500)
        self.seed, self.env_id0 = 0, 0
        self.ep_ret = [0.0] * num_envs
        self.ep_len = [0] * num_envs

    def reset(self, obs_out=None, seed=None):
        if seed is not None:
            self.seed = int(seed)
        obs = np.stack([e.reset(seed=seed)[0] for e in self.envs])
        self.ep_ret = [0.0] * self.n
        self.ep_len = [0] * self.n
        t = self.torch.from_numpy(obs).to(self.device)
        if obs_out is None:
            return t
        obs_out.copy_(t)
        return obs_out

    def step(self, action, obs_out, rew_out, done_out=None, ep_ret_out=None, term_obs_out=None,
             terminated_out=None, truncated_out=None, ep_len_out=None):
        acts = action.tolist()
        obs = np.empty((self.n, self.obs_dim), np.float32)
        tobs = np.empty((self.n, self.obs_dim), np.float32)
        rew = np.empty(self.n, np.float32)
        term = np.zeros(self.n, np.uint8)
        trunc = np.zeros(self.n, np.uint8)
        epr = np.zeros(self.n, np.float32)
        epl = np.zeros(self.n, np.int32)
        for i, e in enumerate(self.envs):
            o, r, te, tr, _ = e.step(acts[i])
            self.ep_ret[i] += r
            self.ep_len[i] += 1
            tobs[i] = o
            rew[i], term[i], trunc[i] = r, te, tr
            epr[i], epl[i] = self.ep_ret[i], self.ep_len[i]
            if te or tr:
                o, _ = e.reset()
                self.ep_ret[i], self.ep_len[i] = 0.0, 0
            obs[i] = o
        T = self.torch
        obs_out.copy_(T.from_numpy(obs).to(self.device))
        rew_out.copy_(T.from_numpy(rew).to(self.device))
        for dst, src in ((done_out, term | trunc), (ep_ret_out, epr), (term_obs_out, tobs),
                         (terminated_out, term), (truncated_out, trunc), (ep_len_out, epl)):
            if dst is not None:
                dst.copy_(T.from_numpy(src).to(self.device))

    def abandon(self, cap, obs_inout, flag_inout=None, ep_ret_out=None, ep_len_out=None):
        """VecEnv.abandon: the scripted episodes (9..21 steps) never reach a trainer's step cap."""
        assert max(self.ep_len) < cap

    def close(self):
        pass
