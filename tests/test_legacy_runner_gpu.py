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
    assert hasattr(agent, "state_norm") and agent.state_norm.running_ms.n == 64 * 12 * 128 + 128
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
