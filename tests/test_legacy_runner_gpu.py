"""utils/ counterparts (runner.train, ReplayBuffer_on_policy / _off_policy, Normalization,
RewardScaling) driven by agents written to the reference's legacy duck-type
(legacy/LunarLander(PPO).py, legacy/CartPole(DQN).py), on lane-parallel envs."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_on_policy_runner_with_legacy_style_agent(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)                                   # ModelLoader writes ./checkpoints/<algo>_<env>.pth
    from torch import nn
    from gymrl_amd.utils.model import ModelLoader
    from torch.distributions import Categorical
    from gymrl_amd import ops
    from gymrl_amd.utils.buffer import ReplayBuffer_on_policy
    from gymrl_amd.utils.runner import BasicConfig, make_env, train, evaluate

    class Config(BasicConfig):
        def __init__(self):
            super().__init__()
            self.env_name, self.algo_name = "CartPole-v1", "PPO"
            self.train_eps, self.num_envs = 10 ** 9, 128
            self.batch_size, self.mini_batch, self.epochs = 128 * 64, 2048, 4
            self.lr = 1e-3

    class PPO(ModelLoader):
        def __init__(self, cfg):
            super().__init__(cfg)
            self.net = nn.Sequential(nn.Linear(cfg.n_states, 64), nn.Tanh()).to(cfg.device)
            self.pi, self.v = nn.Linear(64, cfg.n_actions).to(cfg.device), nn.Linear(64, 1).to(cfg.device)
            params = list(self.net.parameters()) + list(self.pi.parameters()) + list(self.v.parameters())
            self.optimizer = torch.optim.Adam(params, lr=cfg.lr, eps=1e-5)
            self.memory = ReplayBuffer_on_policy(cfg)
            self.learn_step = 0

        @torch.no_grad()
        def choose_action(self, state):
            h = self.net(state)
            act, logp, _, val = ops.categorical_sample(self.pi(h), value=self.v(h).view(-1), seed=1,
                                                       counter=self.learn_step * 100000 + len(self.memory.buffer))
            return act, logp, val

        @torch.no_grad()
        def evaluate(self, state):
            return self.pi(self.net(state)).argmax(-1).to(torch.int32)

        def update(self):
            states, actions, old_probs, adv, v_target = self.memory.sample()
            n = states.shape[0]
            for _ in range(self.cfg.epochs):
                for idx in torch.randperm(n, device=states.device).split(self.cfg.mini_batch):
                    h = self.net(states[idx])
                    dist = Categorical(logits=self.pi(h))
                    ratio = torch.exp(dist.log_prob(actions[idx, 0]) - old_probs[idx, 0])
                    a = adv[idx, 0]
                    ms = torch.min(ratio * a, torch.clamp(ratio, 0.8, 1.2) * a)
                    loss = -torch.where(a < 0, torch.max(ms, 3.0 * a), ms).mean() \
                        + 0.5 * (v_target[idx, 0] - self.v(h).view(-1)).pow(2).mean() - 0.01 * dist.entropy().mean()
                    self.optimizer.zero_grad()
                    loss.backward()
                    self.optimizer.step()
            self.memory.clear()
            self.learn_step += 1
            return {"total_loss": float(loss.item())}

    cfg = Config()
    env = make_env(cfg)
    assert cfg.n_states == 4 and cfg.n_actions == 2 and cfg.max_steps == 500
    agent = PPO(cfg)
    returns, metrics = train(env, agent, cfg, max_vector_steps=64 * 12)
    assert agent.learn_step >= 10 and np.isfinite(metrics["total_loss"])
    # every vector step normalises N observations; every finished episode one more (its reset observation, :107)
    assert hasattr(agent, "state_norm") and agent.state_norm.running_ms.n == 64 * 12 * 128 + 128 + len(returns)
    assert np.mean(returns[-50:]) > np.mean(returns[:50])          # reward-scaled PPO improves on CartPole
    assert evaluate(cfg.env_name, agent, cfg, episodes=8) > 60
    # runner.train() saved through ModelLoader (every save_freq episodes and at the end, utils/runner.py:160-164)
    ck = torch.load(cfg.save_path, weights_only=False)
    assert {"net_state_dict", "pi_state_dict", "v_state_dict", "optimizer_state_dict", "learn_step",
            "state_norm_state_dict", "reward_scaler_state_dict"} <= set(ck) and "memory" not in ck
    fresh = PPO(cfg)
    cfg.load_model = True
    train(make_env(cfg), fresh, cfg, max_vector_steps=1)           # load_model(), then one vector step
    assert fresh.learn_step == agent.learn_step
    assert fresh.state_norm.running_ms.n == agent.state_norm.running_ms.n + 2 * 128      # restored, then reset + 1 step
    for a, b in zip(fresh.pi.parameters(), agent.pi.parameters()):
        assert torch.equal(a, b)
    # test() on a freshly built agent (utils/runner.py:187-206): load_model() parks the normalisation statistics until
    # evaluate() creates the object — the saved running mean / std are applied, not fresh ones
    from gymrl_amd.utils.runner import test as run_test
    fresh2 = PPO(cfg)
    assert not hasattr(fresh2, "state_norm")
    scores = run_test(cfg.env_name, fresh2, cfg)
    assert len(scores) == cfg.test_eps and all(np.isfinite(x) for x in scores)
    assert fresh2.state_norm.running_ms.n == fresh.state_norm.running_ms.n
    # the metrics sink (utils/runner.py:46-49, :101, :145-158): one run directory per train() call with an event file
    import glob
    from gymrl_amd.utils.metrics import read_events
    runs = sorted(glob.glob("exp/PPO_CartPole-v1_*"))
    assert len(runs) >= 1
    ev = read_events(glob.glob(runs[0] + "/events.out.tfevents.*")[0])
    tags = {t for _, _, t, _ in ev}
    assert {"train/total_loss", "train/reward", "train/step"} <= tags
    n_rew = sum(1 for _, _, t, _ in ev if t == "train/reward")
    assert n_rew == len(returns) and sum(1 for _, _, t, _ in ev if t == "train/total_loss") == agent.learn_step


def test_off_policy_buffer_contract():
    from gymrl_amd.utils.buffer import ReplayBuffer_off_policy
    dev = torch.device("cuda:0")

    class Cfg:
        memory_capacity, batch_size, device, seed = 1000, 64, "cuda:0", 3

    buf = ReplayBuffer_off_policy(Cfg())
    for t in range(30):
        s = torch.full((50, 3), float(t), device=dev)
        buf.store((s, torch.full((50, 1), 0.5 * t, device=dev), torch.full((50,), -float(t), device=dev), s + 1,
                   torch.zeros(50, dtype=torch.uint8, device=dev)))
    assert buf.size() == 1000
    s, a, r, s2, d = buf.sample()
    assert s.shape == (64, 3) and a.dtype == torch.float32 and a.shape == (64, 1)
    assert torch.equal(s2, s + 1) and torch.equal(r, -s[:, 0]) and torch.allclose(a[:, 0], 0.5 * s[:, 0])
    assert s.min().item() >= 10.0            # rows 0..9 were overwritten by the ring


@pytest.mark.parametrize("kind", ["on", "off"])
def test_runner_trace_matches_reference(kind):
    """Row H1/U1: the reference utils/runner.py train() on the scripted env (tests/golden/runner_trace.npz) replayed
    by gymrl_amd.utils.runner.train() with num_envs = 1: the order of choose_action / update / save_model calls, every
    stored transition (normalised states incl. the TERMINAL observation as next state, scaled rewards, done /
    terminated flags, log-prob, V and V(next)) and the final running statistics."""
    from conftest import load_golden, rel_close
    from scripted_env import ScriptedVecEnv
    from gymrl_amd.utils.buffer import ReplayBuffer_off_policy, ReplayBuffer_on_policy
    from gymrl_amd.utils.runner import BasicConfig, train
    g = load_golden("runner_trace")
    dev = torch.device("cuda:0")
    cfg = BasicConfig()
    cfg.env_name, cfg.algo_name, cfg.train_eps, cfg.save_freq, cfg.num_envs = "Scripted", "toy", 7, 3, 1
    cfg.max_steps, cfg.batch_size, cfg.gamma, cfg.lamda, cfg.device = 500, 20, 0.99, 0.95, "cuda:0"
    cfg.memory_capacity, cfg.n_states, cfg.n_actions, cfg.seed = 10 ** 6, 8, 4, 0
    calls, stored = [], []

    class Toy:
        def __init__(self):
            self.cfg, self.k, self.learn_step = cfg, 0, 0
            self.memory = ReplayBuffer_on_policy(cfg) if kind == "on" else ReplayBuffer_off_policy(cfg)

        def choose_action(self, state):
            assert state.shape == (1, 8)
            self.k += 1
            calls.append(1)
            a = torch.tensor([(self.k * 5 + 1) % 4], dtype=torch.int32, device=dev)
            if kind == "off":
                return a
            return a, torch.tensor([-0.125 * self.k], device=dev), torch.tensor([0.25 * self.k], device=dev)

        def update(self):
            calls.append(2)
            if kind == "on":
                self.memory.clear()
            self.learn_step += 1
            return {}

        def save_model(self):
            calls.append(3)
    agent = Toy()
    store = agent.memory.store

    def rec(tr):
        stored.append(tuple(x.clone() for x in tr))
        store(tr)
    agent.memory.store = rec
    train(ScriptedVecEnv(1, dev), agent, cfg)
    assert calls == g[kind + "_calls"].tolist()
    assert agent.learn_step == int(g[kind + "_learn_step"])
    col = lambda j: np.stack([t[j].cpu().numpy().reshape(-1) for t in stored]).astype(np.float64)     # noqa: E731
    assert len(stored) == len(g[kind + "_action"])
    assert rel_close(col(0), g[kind + "_state"], 1e-5) <= 1e-5
    assert np.array_equal(col(1).ravel(), g[kind + "_action"])
    assert rel_close(col(2).ravel(), g[kind + "_reward"], 1e-5) <= 1e-5
    if kind == "on":
        assert np.array_equal(col(3).ravel(), g["on_done"]) and np.array_equal(col(4).ravel(), g["on_dw"])
        for j, name in ((5, "log_prob"), (6, "value"), (7, "next_value")):
            assert np.array_equal(col(j).ravel(), g["on_" + name]), name
    else:
        assert rel_close(col(3), g["off_next_state"], 1e-5) <= 1e-5 and np.array_equal(col(4).ravel(), g["off_done"])
    ms = agent.state_norm.running_ms
    assert ms.n == int(g[kind + "_norm"][0])
    assert rel_close(ms.mean.cpu().numpy(), g[kind + "_norm"][1:9], 1e-6) <= 1e-6
    assert rel_close(ms.std.cpu().numpy(), g[kind + "_norm"][9:17], 1e-6) <= 1e-6
    rs = agent.reward_scaler.running_ms
    assert rs.n == int(g[kind + "_rscale"][0]) and rel_close(rs.std.cpu().numpy(), g[kind + "_rscale"][2:3], 1e-6) <= 1e-6


def test_runner_without_a_metrics_sink_and_train_after_evaluate(tmp_path, monkeypatch):
    """cfg.log_metrics = False (no writer: finished episodes must not index the length list that only the sink fills),
    and evaluate() on a fresh agent must not leave a reward scaler sized for the evaluation vector behind."""
    monkeypatch.chdir(tmp_path)
    import glob
    from gymrl_amd.utils.buffer import ReplayBuffer_off_policy
    from gymrl_amd.utils.runner import BasicConfig, make_env, train, evaluate

    class Config(BasicConfig):
        def __init__(self):
            super().__init__()
            self.env_name, self.algo_name = "CartPole-v1", "Rand"
            self.train_eps, self.num_envs, self.batch_size = 10 ** 9, 32, 64
            self.memory_capacity = 4096
            self.log_metrics = False

    class Rand:
        def __init__(self, cfg):
            self.cfg, self.memory, self.k = cfg, ReplayBuffer_off_policy(cfg), 0

        def choose_action(self, state):
            self.k += 1
            return torch.full((state.shape[0],), self.k & 1, dtype=torch.int32, device=state.device)

        evaluate = choose_action

        def update(self):
            return {"loss": 0.0}

    cfg = Config()
    env = make_env(cfg)
    agent = Rand(cfg)
    assert np.isfinite(evaluate(cfg.env_name, agent, cfg, episodes=5))
    assert hasattr(agent, "state_norm") and not hasattr(agent, "reward_scaler")      # evaluation scales no rewards
    returns, _ = train(env, agent, cfg, max_vector_steps=80)                       # alternating actions: ~20-step episodes
    assert len(returns) > 32 and agent.reward_scaler.R.shape[0] == 32
    assert not glob.glob("exp/*")                                                  # no sink, no run directory
    # a scaler left behind for another env count is rebuilt for this vector, its running statistics kept
    from gymrl_amd.utils.normalization import RewardScaling
    n_seen = agent.reward_scaler.running_ms.n
    old = agent.reward_scaler
    agent.reward_scaler = RewardScaling(shape=1, gamma=cfg.gamma, num_envs=5, device=cfg.device)
    agent.reward_scaler.load_state_dict(old.state_dict())
    train(make_env(cfg), agent, cfg, max_vector_steps=2)
    assert agent.reward_scaler.R.shape[0] == 32 and agent.reward_scaler.running_ms.n == n_seen + 2 * 32
