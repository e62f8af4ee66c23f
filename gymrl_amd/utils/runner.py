"""utils/runner.py:16-226 for lane-parallel envs: BasicConfig, make_env, train, evaluate, test,
BenchMark.  The agent duck-type is the reference's (:81-206): `agent.memory` (its class decides
on/off-policy), `agent.choose_action(state)` -> (a, logp, v) on-policy or a off-policy,
`agent.evaluate(state)`, `agent.update()` -> dict, optional `agent.state_norm` /
`agent.reward_scaler`, `agent.save_model()` / `agent.load_model()`, `agent.learn_step`.
States, actions and rewards are [N, ...] device tensors (one vector step per loop turn).
"""
import os
import time

import numpy as np
import torch

from ..envs import EpisodeTracker, VecEnv
from .buffer import ReplayBuffer_on_policy
from .metrics import ScalarWriter, log_monitors
from .normalization import Normalization, RewardScaling


class BasicConfig:
    def __init__(self):
        self.render_mode = 'rgb_array'
        self.train_eps = 500
        self.test_eps = 3
        self.eval_eps = 10
        self.eval_freq = 10
        self.max_steps = 1000
        self.lr_start = 1e-3
        self.lr_end = 1e-5
        self.batch_size = 1024
        self.mini_batch = 16
        self.epochs = 3
        self.clip = 0.2
        self.dual_clip = 3.0
        self.gamma = 0.99
        self.lamda = 0.95
        self.val_coef = 0.5
        self.ent_coef = 1e-2
        self.grad_clip = 0.5
        self.load_model = False
        self.save_freq = 50
        self.use_state_norm = True
        self.use_reward_scale = True
        self.device = "cuda"
        self.num_envs = 64

    def show(self):
        print('-' * 30 + '参数列表' + '-' * 30)
        for k, v in vars(self).items():
            print(k, '=', v)
        print('-' * 60)


def make_env(cfg):
    """:52-78: builds the (vectorised) env and fills cfg.n_states / n_actions / action_bound / max_steps."""
    env = VecEnv(cfg.env_name, cfg.num_envs, device=cfg.device, seed=getattr(cfg, "seed", 0) or 0)
    cfg.state_shape = env.observation_space.shape
    cfg.n_states = env.obs_dim
    if env.discrete:
        cfg.n_actions, cfg.action_bound = env.act_dim, None
    else:
        cfg.n_actions, cfg.action_bound = env.act_dim, float(env.action_space.high[0])
    cfg.max_steps = env.max_steps
    return env


def ensure_normalizers(agent, cfg, num_envs, reward_scaler=True):
    """Create `agent.state_norm` / `agent.reward_scaler` when the config asks for them and the agent has none yet, and
    hand them the statistics a checkpoint carried (ModelLoader.load_model parks them in `agent._pending_state` when the
    object does not exist at load time).  The reference pickles the objects themselves (utils/model.py:337-366), so its
    test() path finds them after load_model(); here train(), evaluate() and test() all go through this."""
    if getattr(cfg, "use_state_norm", False) and not hasattr(agent, 'state_norm'):
        agent.state_norm = Normalization(shape=cfg.n_states, device=cfg.device)
    if reward_scaler and getattr(cfg, "use_reward_scale", False):
        # per-env running return R [num_envs]: one left behind for another env count (e.g. by an evaluation vector) is
        # rebuilt; its running statistics travel through the state dict, which tolerates the new R shape
        old = getattr(agent, 'reward_scaler', None)
        if old is None or int(old.R.shape[0]) != int(num_envs):
            agent.reward_scaler = RewardScaling(shape=1, gamma=cfg.gamma, num_envs=num_envs, device=cfg.device)
            if old is not None:
                agent.reward_scaler.load_state_dict(old.state_dict())
    pending = getattr(agent, "_pending_state", {})
    for attr in ("state_norm", "reward_scaler"):
        if attr in pending and hasattr(agent, attr):
            getattr(agent, attr).load_state_dict(pending.pop(attr))


def make_writer(cfg):
    """:101 — `./exp/<algo>_<env>_<timestamp>` (cfg.log_dir overrides the root; cfg.log_metrics = False: no sink)."""
    if not getattr(cfg, "log_metrics", True):
        return None
    root = getattr(cfg, "log_dir", "./exp")
    stamp = time.strftime("%Y%m%d-%H%M%S")
    logdir = f'{root}/{getattr(cfg, "algo_name", "agent")}_{cfg.env_name.replace("/", "-")}_{stamp}'
    k = 0
    while os.path.exists(logdir if k == 0 else f"{logdir}-{k}"):      # two train() calls within one second: separate runs
        k += 1
    return ScalarWriter(logdir if k == 0 else f"{logdir}-{k}")


def train(env, agent, cfg, max_vector_steps=None):
    """:81-166.  Returns (finished-episode returns, the last update()'s metrics dict)."""
    if cfg.load_model:
        agent.load_model()
    ensure_normalizers(agent, cfg, env.n)
    writer = make_writer(cfg)
    on_policy = isinstance(agent.memory, ReplayBuffer_on_policy)
    N, D, dev = env.n, env.obs_dim, env.device
    obs, nxt, tobs = (torch.empty(N, D, device=dev) for _ in range(3))
    rew = torch.empty(N, device=dev)
    term = torch.zeros(N, dtype=torch.uint8, device=dev)
    trunc = torch.zeros(N, dtype=torch.uint8, device=dev)
    returns = []
    tracker = EpisodeTracker(N, dev)
    norm = (lambda x, **k: agent.state_norm(x, **k)) if cfg.use_state_norm else (lambda x, **k: x.clone())
    env.reset(obs, seed=int(np.random.randint(1, 2 ** 31 - 1)))
    state = norm(obs)                                                                   # :107
    if on_policy:
        action, log_prob, value = agent.choose_action(state)                            # :109-110
    else:
        action = agent.choose_action(state)                                             # :112
    step, limit = 0, max_vector_steps or (cfg.train_eps * cfg.max_steps // N + 1)
    metrics, saved, episodes = {}, 0, 0
    while episodes < cfg.train_eps and step < limit:
        ep_ret, done = tracker.slot()
        env.step(action, nxt, rew, done_out=done, term_obs_out=tobs, ep_ret_out=ep_ret, terminated_out=term,
                 truncated_out=trunc, ep_len_out=tracker.len_slot() if writer is not None else None)
        r = agent.reward_scaler(rew, done) if cfg.use_reward_scale else rew               # :121 (R zeroed after use where done = reset() :99)
        next_state = norm(tobs)                              # :122 — the TERMINAL observation where an episode ended
        if on_policy:
            next_action, next_log_prob, next_value = agent.choose_action(next_state)     # chosen before storing (:125)
            agent.memory.store((state, action, r.clone(), done.clone(), term.clone(), log_prob, value, next_value))
            action, log_prob, value = next_action, next_log_prob, next_value
        else:
            agent.memory.store((state, action, r.clone(), next_state, done.clone()))     # :133
            action = agent.choose_action(next_state)                                     # :134, before the update
        state = next_state
        step += 1
        if agent.memory.size() >= cfg.batch_size:                                        # :138-140
            metrics = agent.update()
            log_monitors(writer, metrics, agent, 'train', getattr(agent, "learn_step", step))   # :146-147
        # envs whose episode ended start a new one (:97-112 of the next loop turn): normalise the reset observation
        # (a second statistics update, as in the reference) and choose its first action.  One small host read per
        # vector step; this runner is the legacy contract, not the throughput path.
        n_before = len(returns)
        tracker.advance(returns)                 # drains finished episodes every 16 vector steps (no per-step host sync)
        if writer is not None:                   # (episode lengths are only tracked for the sink: cfg.log_metrics = False)
            for k in range(n_before, len(returns)):                                      # :157, per finished episode
                log_monitors(writer, {'reward': returns[k], 'step': tracker.lengths[k]}, agent, 'train', k)
        if bool(done.any()):
            idx = done.nonzero().view(-1)
            period = cfg.eval_freq * N            # :160-162 every eval_freq episodes of ONE env: eval_freq * N of the vector
            if (episodes + idx.numel()) // period > episodes // period and writer is not None:
                log_monitors(writer, {'reward': evaluate(cfg.env_name, agent, cfg)}, agent, 'eval', episodes + idx.numel())
            episodes += idx.numel()
            if hasattr(agent, "save_model") and episodes // cfg.save_freq > saved:       # :160-161, every save_freq episodes
                saved = episodes // cfg.save_freq
                agent.save_model()
            if episodes >= cfg.train_eps:
                break
            fresh = norm(nxt.index_select(0, idx))
            state = state.clone()
            state[idx] = fresh
            if on_policy:
                a2, l2, v2 = agent.choose_action(fresh)
                action, log_prob, value = action.clone(), log_prob.clone(), value.clone()
                action[idx], log_prob[idx], value[idx] = a2, l2, v2
            else:
                action = action.clone()
                action[idx] = agent.choose_action(fresh)
    n_before = len(returns)
    tracker.flush(returns)
    if writer is not None:
        for k in range(n_before, len(returns)):
            log_monitors(writer, {'reward': returns[k], 'step': tracker.lengths[k]}, agent, 'train', k)
    if hasattr(agent, "save_model"):                                                   # :164
        agent.save_model()
    if writer is not None:
        writer.close()                                                                 # :166
    return returns, metrics


@torch.no_grad()
def evaluate(env_name, agent, cfg, episodes=None):
    """:169-184: deterministic episodes on a fresh env vector; mean return."""
    n = episodes or cfg.eval_eps
    # a freshly built agent after load_model(): statistics from the checkpoint.  Only the state normaliser — evaluation
    # never scales rewards, and a scaler sized for the n evaluation episodes must not be what a later train() finds
    ensure_normalizers(agent, cfg, n, reward_scaler=False)
    env = VecEnv(env_name, n, device=cfg.device, seed=12345, env_id0=1 << 40)
    obs = env.reset()
    nxt, rew = torch.empty_like(obs), torch.empty(n, device=env.device)
    done = torch.zeros(n, dtype=torch.uint8, device=env.device)
    ep_ret = torch.zeros(n, device=env.device)
    result = torch.full((n,), float("nan"), device=env.device)
    for _ in range(env.max_steps + 1):
        state = agent.state_norm(obs, update=False) if getattr(cfg, "use_state_norm", False) else obs
        env.step(agent.evaluate(state), nxt, rew, done_out=done, ep_ret_out=ep_ret)
        result = torch.where(done.bool() & torch.isnan(result), ep_ret, result)
        obs, nxt = nxt, obs
        if not torch.isnan(result).any():
            break
    return float(result.mean().item())


def test(env_name, agent, cfg):
    """:187-206 (no renderer on the batched env)."""
    agent.load_model()
    return [evaluate(env_name, agent, cfg, episodes=1) for _ in range(cfg.test_eps)]


class BenchMark:
    """:209-226."""

    @staticmethod
    def train(algo, config, max_vector_steps=None):
        cfg = config()
        env = make_env(cfg)                      # must precede agent construction: it fills the dims
        agent = algo(cfg)
        t0 = time.time()
        out = train(env, agent, cfg, max_vector_steps)
        print(f"train: {time.time() - t0:.1f}s")
        return agent, out

    @staticmethod
    def test(algo, config):
        cfg = config()
        cfg.render_mode = 'human'
        make_env(cfg)
        agent = algo(cfg)
        return test(cfg.env_name, agent, cfg)
