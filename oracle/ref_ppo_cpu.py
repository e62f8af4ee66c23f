"""CPU counterpart of the reference's algorithms/ppo_lunarlander.py training loop.
TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests/).

The GPU box never receives reference source, so the "reference CPU path timed beside
the MI355X numbers"  is this structurally faithful port:
  * ONE env instance (the oracle's C LunarLander restatement), stepped one transition
    at a time (ppo_lunarlander.py:198-231): B=1 ActorCritic forward, Categorical
    sample, three host scalar reads, python-list buffer appends, reset at every
    rollout start (:200);
  * python-loop float64 GAE (:179-196), population-std advantage normalisation (:236);
  * 10 epochs x 32 minibatches of 64 (:252-307) with np.random.shuffle, autograd
    loss (:278-300), clip_grad_norm_(0.5), torch.optim.Adam(lr, eps=1e-5), 5 .item()
    metrics per minibatch (:309-322), LR annealing (:337-341).
Everything runs on the host with torch CPU kernels.
"""
import time
from collections import deque

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical

from . import oracle as orc


class Config:
    def __init__(self):
        self.env_name = "LunarLander-v3"
        self.seed = 0
        self.max_train_steps = 1_000_000
        self.update_freq = 2048
        self.num_epochs = 10
        self.batch_size = 64
        self.gamma = 0.99
        self.gae_lambda = 0.95
        self.clip_eps = 0.2
        self.dual_clip = 3.0
        self.entropy_coef = 0.01
        self.value_coef = 0.5
        self.max_grad_norm = 0.5
        self.lr = 3e-4
        self.anneal_lr = True
        self.hidden_dim = 256


def _layer_init(layer, std=np.sqrt(2)):
    nn.init.orthogonal_(layer.weight, gain=std)
    nn.init.constant_(layer.bias, 0)
    return layer


class ActorCritic(nn.Module):
    def __init__(self, state_dim, action_dim, hidden_dim=256):
        super().__init__()
        self.shared = nn.Sequential(_layer_init(nn.Linear(state_dim, hidden_dim)), nn.Tanh(),
                                    _layer_init(nn.Linear(hidden_dim, hidden_dim)), nn.Tanh())
        self.actor = nn.Sequential(_layer_init(nn.Linear(hidden_dim, hidden_dim)), nn.Tanh(),
                                   _layer_init(nn.Linear(hidden_dim, action_dim), std=0.01))
        self.critic = nn.Sequential(_layer_init(nn.Linear(hidden_dim, hidden_dim)), nn.Tanh(),
                                    _layer_init(nn.Linear(hidden_dim, 1), std=1.0))

    def forward(self, x):
        f = self.shared(x)
        return self.actor(f), self.critic(f)


class OneEnv:
    """gymnasium-style single env on top of the oracle's vector API (n = 1)."""

    def __init__(self, seed):
        self.kind, self.seed = orc.LUNARLANDER, seed
        self._mk(seed)

    def _mk(self, seed):
        self.env = orc.Env(self.kind, 1, seed=seed)

    def reset(self, seed=None):
        if seed is not None:
            self._mk(seed)
        return self.env.reset()[0], {}

    def step(self, action):
        r = self.env.step(np.array([action], np.int32))
        # the oracle auto-resets; hand back the pre-reset observation like gymnasium does
        self._next_reset_obs = r["obs"][0]
        return r["term_obs"][0], float(r["rew"][0]), bool(r["terminated"][0]), bool(r["truncated"][0]), {}

    def reset_after_done(self):
        return self._next_reset_obs, {}


class PPOTrainerCPU:
    def __init__(self, config):
        self.cfg = config
        torch.manual_seed(config.seed)
        np.random.seed(config.seed)
        self.env = OneEnv(config.seed)
        self.model = ActorCritic(8, 4, config.hidden_dim)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=config.lr, eps=1e-5)
        self.buf = dict(states=[], actions=[], log_probs=[], values=[], rewards=[], dones=[])
        self.step_count = 0
        self.episode_rewards = deque(maxlen=100)

    def compute_gae(self, next_value):
        """GAE(gamma, lambda) backwards over the one env's rollout, in the precision the reference computes it in: rewards and
        values collected as python floats (float64 arrays), the episode-end mask float32 (ppo_lunarlander.py:179-196)."""
        g, lam = self.cfg.gamma, self.cfg.gae_lambda
        r = np.asarray(self.buf["rewards"])
        v = np.asarray(self.buf["values"] + [next_value])
        alive = 1 - np.asarray(self.buf["dones"], dtype=np.float32)       # 0 where the episode ended at step t
        adv = np.zeros_like(r)
        carry, t = 0.0, len(r)
        while t > 0:
            t -= 1
            td = r[t] + g * v[t + 1] * alive[t] - v[t]                  # one-step TD error
            carry = td + g * lam * alive[t] * carry
            adv[t] = carry
        return adv, adv + v[:-1]

    def collect_rollout(self):
        for v in self.buf.values():
            v.clear()
        state, _ = self.env.reset(seed=self.cfg.seed + self.step_count)
        episode_reward = 0.0
        for _ in range(self.cfg.update_freq):
            st = torch.tensor(state, dtype=torch.float32).unsqueeze(0)
            with torch.no_grad():
                logits, value = self.model(st)
                dist = Categorical(logits=logits)
                action = dist.sample()
                a, lp, v = action.item(), dist.log_prob(action).item(), value.squeeze().item()
            next_state, reward, terminated, truncated, _ = self.env.step(a)
            done = terminated or truncated
            for k, x in zip(self.buf, (state, a, lp, v, reward, done)):
                self.buf[k].append(x)
            state = next_state
            episode_reward += reward
            self.step_count += 1
            if done:
                self.episode_rewards.append(episode_reward)
                state, _ = self.env.reset_after_done()
                episode_reward = 0.0
        with torch.no_grad():
            return self.model(torch.tensor(state, dtype=torch.float32).unsqueeze(0))[1].squeeze().item()

    def update(self, next_value):
        cfg = self.cfg
        advantages, returns = self.compute_gae(next_value)
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
        states = torch.tensor(np.array(self.buf["states"]), dtype=torch.float32)
        actions = torch.tensor(self.buf["actions"], dtype=torch.long)
        old_lp = torch.tensor(self.buf["log_probs"], dtype=torch.float32)
        adv_t = torch.tensor(advantages, dtype=torch.float32)
        ret_t = torch.tensor(returns, dtype=torch.float32)
        n = len(self.buf["states"])
        indices = np.arange(n)
        out = [[], [], [], [], []]
        for _ in range(cfg.num_epochs):
            np.random.shuffle(indices)
            for start in range(0, n, cfg.batch_size):
                mb = indices[start:start + cfg.batch_size]
                logits, values = self.model(states[mb])
                dist = Categorical(logits=logits)
                new_lp, entropy = dist.log_prob(actions[mb]), dist.entropy()
                ratio = torch.exp(new_lp - old_lp[mb])
                surr1 = ratio * adv_t[mb]
                surr2 = torch.clamp(ratio, 1 - cfg.clip_eps, 1 + cfg.clip_eps) * adv_t[mb]
                min_surr = torch.min(surr1, surr2)
                policy_loss = -torch.mean(torch.where(adv_t[mb] < 0, torch.max(min_surr, cfg.dual_clip * adv_t[mb]), min_surr))
                value_loss = cfg.value_coef * torch.mean((values.squeeze(-1) - ret_t[mb]).pow(2))
                loss = policy_loss + value_loss - cfg.entropy_coef * entropy.mean()
                self.optimizer.zero_grad()
                loss.backward()
                nn.utils.clip_grad_norm_(self.model.parameters(), cfg.max_grad_norm)
                self.optimizer.step()
                out[0].append(policy_loss.item()); out[1].append(value_loss.item()); out[2].append(entropy.mean().item())
                with torch.no_grad():
                    out[3].append(((ratio < 1 - cfg.clip_eps) | (ratio > 1 + cfg.clip_eps)).float().mean().item())
                    out[4].append((old_lp[mb] - new_lp).mean().item())
        return dict(zip(("policy_loss", "value_loss", "entropy", "clip_frac", "approx_kl"), (np.mean(o) for o in out)))

    def iteration(self):
        if self.cfg.anneal_lr:
            lr = self.cfg.lr * (1.0 - self.step_count / self.cfg.max_train_steps)
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        return self.update(self.collect_rollout())


def _time_once(budget_s, update_freq, threads):
    torch.set_num_threads(threads)
    cfg = Config()
    tr = PPOTrainerCPU(cfg)
    cfg.update_freq = 256                 # warm-up: page in torch kernels
    tr.iteration()
    cfg.update_freq = update_freq
    tr.step_count = 0
    t0 = time.perf_counter()
    cycles, roll_s = 0, 0.0
    while True:
        r0 = time.perf_counter()
        nv = tr.collect_rollout()
        roll_s += time.perf_counter() - r0
        tr.update(nv)
        cycles += 1
        if time.perf_counter() - t0 >= budget_s or cycles >= 8:
            break
    dt = time.perf_counter() - t0
    steps = cycles * update_freq
    return steps / dt, steps / roll_s, cycles, dt


def time_cpu_baseline(budget_s=20.0, update_freq=2048):
    """Time the loop above on the host cores, once with torch's default thread count and once
    single-threaded (B = 1 GEMVs over-thread badly); report the faster.  Returns bench.py's
    cpu_baseline dict."""
    import os
    default_threads = torch.get_num_threads()
    runs = []
    for threads in sorted({1, min(default_threads, 8), default_threads}):
        v, roll, cycles, dt = _time_once(budget_s / 3.0, update_freq, threads)
        runs.append((v, threads, roll, cycles, dt))
    torch.set_num_threads(default_threads)
    best = max(runs)
    detail = "; ".join(f"{t} thr: {v:.0f} steps/s ({c} cycles, {d:.1f}s)" for v, t, _, c, d in runs)
    return dict(value=best[0], unit="env-steps/s", cores=best[1], kind="port", host_cores=os.cpu_count(),
                sample=f"rollout+update cycles of {update_freq} steps, 1 env (oracle C LunarLander), B=1 torch-CPU "
                       f"policy forward per step, 10 epochs x {update_freq // 64} minibatches of 64; {detail}; "
                       f"rollout-only {best[2]:.0f} steps/s at {best[1]} threads")


if __name__ == "__main__":
    print(time_cpu_baseline(10.0))
