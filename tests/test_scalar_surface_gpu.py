"""The reference's SCALAR surface at num_envs == 1: each trainer is driven exactly like the body of the
reference's own train() loop — gymnasium-style env calls, numpy observations in, python int / float / np.ndarray out,
python scalars pushed into the replay buffer — and must behave like the reference's objects do."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_dqn_loop_written_against_the_reference():
    """dqn_cartpole.py:170-212 verbatim in shape: select_action(np state) -> int, env.step(int), memory.push(host scalars),
    update() -> float."""
    from gymrl_amd.dqn_cartpole import Config, DQNTrainer
    cfg = Config()
    cfg.num_envs, cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.seed = 1, 32, 16, 512, 3
    tr = DQNTrainer(cfg)
    env = tr.env.gym
    losses, returns = [], []
    for episode in range(6):
        state, _ = env.reset()
        assert isinstance(state, np.ndarray) and state.shape == (4,) and state.dtype == np.float32
        episode_reward = 0.0
        for step in range(60):
            action = tr.select_action(state)
            assert type(action) is int and action in (0, 1)
            next_state, reward, terminated, truncated, _ = env.step(action)
            assert type(reward) is float and type(terminated) is bool and type(truncated) is bool
            done = terminated or truncated
            tr.memory.push(state, action, reward, next_state, done)
            loss = tr.update()
            assert type(loss) is float
            losses.append(loss)
            state = next_state
            episode_reward += reward
            if done:
                break
        returns.append(episode_reward)
    assert len(tr.memory) == sum(int(r) for r in returns) and len(tr.memory) >= cfg.batch_size
    assert losses[0] == 0.0 and any(v > 0 for v in losses)            # 0.0 while warming up (:136-137), then real losses
    # the terminal observation is what step() returned and what was stored (dqn_cartpole.py:183), not the reset observation
    ring_next = tr.memory.ring[3][:len(tr.memory)].cpu().numpy()
    flags = tr.memory.ring[4][:len(tr.memory)].cpu().numpy().astype(bool)
    assert flags.sum() >= 1 and np.all((np.abs(ring_next[flags][:, 0]) > 2.4) | (np.abs(ring_next[flags][:, 2]) > 0.2095))
    # host observation in == device observation in
    s = np.array([0.01, -0.02, 0.03, 0.04], np.float32)
    a_host = tr.select_action(s, deterministic=True)
    a_dev = tr.select_action(torch.from_numpy(s[None]).to(tr.device), deterministic=True)
    assert type(a_host) is int and torch.is_tensor(a_dev) and a_host == int(a_dev[0])


def test_rainbow_loop_written_against_the_reference():
    """rainbow_dqn_cartpole.py:363-405: select_action -> int, store_transition(host scalars incl. terminal), update() -> float."""
    from gymrl_amd.rainbow_dqn_cartpole import Config, RainbowDQNTrainer
    cfg = Config()
    cfg.num_envs, cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.seed = 1, 32, 16, 4096, 4
    tr = RainbowDQNTrainer(cfg)
    env = tr.env.gym
    n_steps, losses = 0, []
    for episode in range(5):
        state, _ = env.reset()
        for step in range(tr.max_steps_per_episode):
            action = tr.select_action(state)
            assert type(action) is int
            next_state, reward, terminated, truncated, _ = env.step(action)
            done = terminated or truncated
            terminal = done and step != tr.max_steps_per_episode - 1              # :376
            tr.memory.store_transition(state, action, reward, next_state, terminal, done)
            losses.append(tr.update())
            state = next_state
            n_steps += 1
            if done:
                break
    assert tr.total_steps == n_steps and len(tr.memory) == n_steps - (cfg.n_steps - 1)
    assert all(type(v) is float for v in losses) and any(v > 0 for v in losses)
    tree = tr.memory.sum_tree.tree.cpu().numpy()
    assert tree[0] > 0 and abs(tree[0] - tree[cfg.memory_capacity - 1:].sum()) <= 1e-9 * tree[0]


def test_sac_loop_written_against_the_reference():
    """sac_pendulum.py:269-310: select_action(np state) -> np.ndarray [act_dim], env.step(ndarray), push, update() -> 3 floats."""
    from gymrl_amd.sac_pendulum import Config, SACTrainer
    cfg = Config()
    cfg.num_envs, cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.seed = 1, 32, 16, 512, 5
    tr = SACTrainer(cfg)
    env = tr.env.gym
    state, _ = env.reset()
    assert state.shape == (3,)
    out = []
    for step in range(230):                                                        # crosses Pendulum's 200-step time limit
        action = tr.select_action(state)
        assert isinstance(action, np.ndarray) and action.shape == (1,) and abs(float(action[0])) <= 2.0
        next_state, reward, terminated, truncated, _ = env.step(action)
        done = terminated or truncated
        assert not terminated and truncated == (step == 199)
        tr.memory.push(state, action, reward, next_state, done)
        out.append(tr.update())
        state = next_state
        if done:
            state, _ = env.reset()                                                 # the episode the kernel already started
    assert len(tr.memory) == 230
    assert all(len(o) == 3 and all(type(v) is float for v in o) for o in out) and any(o[1] > 0 for o in out)
    assert abs(float(np.hypot(state[0], state[1])) - 1.0) < 1e-6                   # (cos, sin, thdot)
    mean_action = tr.select_action(state, deterministic=True)
    assert isinstance(mean_action, np.ndarray) and mean_action.shape == (1,)


def test_ppo_surface_at_one_env():
    """ppo_lunarlander.py:179-231,:332-345: collect_rollout() -> float, compute_gae(float) -> two float64 arrays [T],
    update(float) -> the five-metric dict of python floats."""
    from gymrl_amd.ppo_lunarlander import Config, PPOTrainer
    cfg = Config()
    cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.hidden_dim, cfg.seed = 1, 128, 2, 32, 64, 6
    tr = PPOTrainer(cfg)
    next_value = tr.collect_rollout()
    assert type(next_value) is float
    adv, ret = tr.compute_gae(next_value)
    assert isinstance(adv, np.ndarray) and adv.shape == (128,) and adv.dtype == np.float64 and ret.shape == (128,)
    # the same numbers as the engine path on the device slab (float32 storage)
    adv_d, ret_d = tr.compute_gae()
    assert np.array_equal(adv.astype(np.float32), adv_d[:, 0].cpu().numpy()) and np.array_equal(ret.astype(np.float32), ret_d[:, 0].cpu().numpy())
    # returns = advantages + values (:195)
    assert np.allclose(ret - adv, tr.buffer.values[:, 0].cpu().numpy(), atol=1e-5)
    metrics = tr.update(next_value)
    assert set(metrics) == {"policy_loss", "value_loss", "entropy", "clip_frac", "approx_kl"}
    assert all(isinstance(v, float) and np.isfinite(v) for v in metrics.values())      # np.mean's float64, as in the reference (:323-329)
    assert tr.step_count == 128 and isinstance(tr.episode_rewards, type(tr.episode_rewards)) and tr.episode_rewards.maxlen == 100
    # engine path unchanged: N > 1 keeps device tensors
    cfg2 = Config()
    cfg2.num_envs, cfg2.update_freq, cfg2.num_epochs, cfg2.num_minibatches, cfg2.hidden_dim, cfg2.seed = 64, 16, 1, 2, 64, 6
    tr2 = PPOTrainer(cfg2)
    nv = tr2.collect_rollout()
    assert torch.is_tensor(nv) and nv.shape == (64,)


def test_ppo_full_update_model_takes_numpy():
    """ppo_full_lunarlander.py:681-700: compute_advantages() feeds update_model(advantages, returns); the reference hands
    numpy arrays over (:537-556) — the same update either way."""
    from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer
    outs = []
    for as_numpy in (False, True):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed = 1, 64, 1, 32, 7
        tr = PPOTrainer(cfg)
        tr.collect_experience()
        adv, ret = tr.compute_advantages()
        if as_numpy:
            adv, ret = adv.cpu().numpy()[:, 0], ret.cpu().numpy()[:, 0]
            assert adv.shape == (64,)
        m = tr.update_model(adv, ret)
        outs.append((m, tr.flat_params.clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0] == outs[1][0]


def test_script_entry_point_overrides_and_sigint(monkeypatch):
    """`python -m gymrl_amd.<algorithm>` = the reference's __main__ block (ppo_lunarlander.py:429-445): Config, trainer, a
    SIGINT handler that runs test() and exits, train(), test()."""
    import signal
    from gymrl_amd.dqn_cartpole import Config, DQNTrainer
    from gymrl_amd.utils import cli
    calls = []
    monkeypatch.setattr(DQNTrainer, "train", lambda self: calls.append("train"))
    monkeypatch.setattr(DQNTrainer, "test", lambda self: calls.append("test"))
    old = signal.getsignal(signal.SIGINT)
    try:
        tr = cli.run_script(Config, DQNTrainer, ["--num_envs", "8", "--hidden_dim", "32", "--seed", "1"])
        assert calls == ["train", "test"] and tr.cfg.num_envs == 8 and tr.cfg.hidden_dim == 32
        handler = signal.getsignal(signal.SIGINT)
        assert callable(handler) and handler is not old
        with pytest.raises(SystemExit) as e:
            handler(signal.SIGINT, None)                                            # Ctrl+C: test(), then exit 0
        assert e.value.code == 0 and calls == ["train", "test", "test"]
    finally:
        signal.signal(signal.SIGINT, old)
    with pytest.raises(SystemExit):
        cli.apply_overrides(Config(), ["--no_such_attribute", "1"])
