"""The hipGraph-replayed update loops reproduce the eager path bit for bit (gymrl_amd/graphs.py)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(mod, cls, graphs, steps, setup=None):
    cfg = mod.Config()
    cfg.num_envs, cfg.max_episodes, cfg.batch_size, cfg.seed, cfg.use_graphs = 64, 10**9, 128, 5, graphs
    if setup:
        setup(cfg)
    torch.manual_seed(11)
    tr = getattr(mod, cls)(cfg)
    tr.train(max_vector_steps=steps)
    torch.cuda.synchronize()
    return tr


def test_store_scalars_and_device_bias_adam():
    """gymrl_store_scalars round trip; gymrl_adam_step with the device-resident bias block == host-step form."""
    from gymrl_amd import ops
    dev = torch.device("cuda:0")
    blk = torch.zeros(64, dtype=torch.float32, device=dev)
    payload = np.arange(64, dtype=np.float32).tobytes()
    ops.store_scalars(blk, payload)
    assert np.array_equal(blk.cpu().numpy(), np.arange(64, dtype=np.float32))
    ops.store_scalars(blk, np.array([7.5], np.float32).tobytes())
    assert blk[0].item() == 7.5 and blk[1].item() == 1.0
    rng = np.random.default_rng(0)
    p0, g0 = rng.normal(size=1003).astype(np.float32), rng.normal(size=1003).astype(np.float32)
    outs = []
    for use_dev in (False, True):
        p, g = torch.from_numpy(p0).to(dev), torch.from_numpy(g0).to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        bias = torch.zeros(4, device=dev)
        for step in range(1, 6):
            g.copy_(torch.from_numpy(g0 * step).to(dev))
            if use_dev:
                ops.store_scalars(bias, ops.adam_bias(3e-4, 0.9, 0.999, step))
                ops.adam_step(p, g, m, v, 3e-4, 0.9, 0.999, 1e-8, 1, bias_dev=bias)
            else:
                ops.adam_step(p, g, m, v, 3e-4, 0.9, 0.999, 1e-8, step)
        outs.append((p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_sac_graphed_update_equals_eager():
    """The LAYER-BY-LAYER update as a per-update hipGraph (fused_step off: the fused step is 5 launches and replays whole
    chunks of vector steps instead — tests/test_fused_step_gpu.py, tests/test_step_chunk_gpu.py)."""
    from gymrl_amd import sac_pendulum

    def layerwise(cfg):
        cfg.fused_step = False
    eager = _run(sac_pendulum, "SACTrainer", False, 40, layerwise)
    graph = _run(sac_pendulum, "SACTrainer", True, 40, layerwise)
    assert graph._graph is not None and graph._graph.graph is not None          # captured and replayed
    assert graph.critic_optimizer.step_count == eager.critic_optimizer.step_count > 30
    for name in ("actor_flat", "critic_flat", "critic_target_flat", "log_alpha", "_alpha_m", "_alpha_v"):
        assert torch.equal(getattr(eager, name), getattr(graph, name)), name
    assert torch.equal(eager.actor_optimizer.m, graph.actor_optimizer.m)


def test_dqn_graphed_update_equals_eager():
    from gymrl_amd import dqn_cartpole
    eager = _run(dqn_cartpole, "DQNTrainer", False, 60)
    graph = _run(dqn_cartpole, "DQNTrainer", True, 60)
    assert graph._graph is not None and graph._graph.graph is not None
    assert graph.optimizer.step_count == eager.optimizer.step_count > 40
    assert torch.equal(eager.flat_params, graph.flat_params) and torch.equal(eager.optimizer.v, graph.optimizer.v)
    assert list(eager.episode_rewards) == list(graph.episode_rewards)


def test_rainbow_graphed_update_equals_eager():
    """Incl. the PER tree (priorities written inside the graph), the staged NoisyNet draws and the annealed lr."""
    from gymrl_amd import rainbow_dqn_cartpole as rb

    def setup(cfg):
        cfg.memory_capacity = 1 << 14
    outs = []
    for graphs in (False, True):
        rb.NoisyLinear._counter = 0
        outs.append(_run(rb, "RainbowDQNTrainer", graphs, 60, setup))
    eager, graph = outs
    assert graph._graph is not None and graph._graph.graph is not None
    assert graph.optimizer.step_count == eager.optimizer.step_count > 40
    assert torch.equal(eager.flat_params, graph.flat_params) and torch.equal(eager.target_flat, graph.target_flat)
    assert torch.equal(eager.memory.sum_tree.tree, graph.memory.sum_tree.tree)
    assert eager.optimizer.param_groups[0]["lr"] == graph.optimizer.param_groups[0]["lr"]


def test_ppo_full_graphed_minibatch_equals_eager():
    """PPO-full: the minibatch body (mHC forward, L3 loss, backward, clip + Adam) captured per update_model() call and
    replayed == the eager loop, over three iterations (entropy coefficient and lr annealed between the captures)."""
    from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer

    def run(graphs):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed, cfg.mhc_dim = 64, 32, 2, 256, 1, 32
        cfg.use_graphs = graphs
        torch.manual_seed(5)
        tr = PPOTrainer(cfg)
        ms = []
        for _ in range(3):
            tr.collect_experience()
            adv, ret = tr.compute_advantages()
            ms.append(tr.update_model(adv, ret))
        return tr, ms
    (a, ma), (b, mb) = run(False), run(True)
    assert b._g_idx is not None and b.optimizer.step_count == a.optimizer.step_count == 3 * 2 * 8
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.optimizer.v, b.optimizer.v)
    for x, y in zip(ma, mb):
        assert x == y


def test_ppo_lstm_graphed_equals_eager():
    """Recurrent PPO: rollout forward (GRU state in / out) and the sequence-minibatch body (GRU unrolled over the
    window, masked loss with its count pass, RND loss, backward, clip + Adam) replayed as hipGraphs == eager."""
    from gymrl_amd.ppo_lstm_lunarlander import Config, PPOTrainer

    def run(graphs):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.seq_len, cfg.batch_size, cfg.num_epochs, cfg.seed = 32, 64, 8, 64, 2, 2
        cfg.mhc_dim, cfg.rnn_hidden, cfg.head_hidden, cfg.rnd_embed, cfg.use_graphs = 32, 64, 64, 64, graphs
        torch.manual_seed(6)
        tr = PPOTrainer(cfg)
        ms = []
        for _ in range(3):
            tr.collect_experience()
            adv, ret = tr.compute_advantages()
            ms.append(tr.update_model(adv, ret))
        return tr, ms
    (a, ma), (b, mb) = run(False), run(True)
    assert b._g_seq is not None and b._fwd_graph is not None
    assert b.optimizer.step_count == a.optimizer.step_count == 3 * 2 * 4
    assert torch.equal(a.buffer.hidden_states, b.buffer.hidden_states) and torch.equal(a.buffer.rewards, b.buffer.rewards)
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.optimizer.v, b.optimizer.v)
    for x, y in zip(ma, mb):
        assert x == y


@pytest.mark.parametrize("algo", ["ddpg", "dsac"])
def test_ddpg_dsac_graphed_update_equals_eager(algo):
    from gymrl_amd import ddpg_pendulum, sac_cartpole
    mod, cls = {"ddpg": (ddpg_pendulum, "DDPGTrainer"), "dsac": (sac_cartpole, "SACTrainer")}[algo]
    eager = _run(mod, cls, False, 40)
    graph = _run(mod, cls, True, 40)
    assert graph._graph is not None and graph._graph.step_fn.graph is not None
    for name in ("actor_flat",) + (("critic_flat", "critic_target_flat", "actor_target_flat") if algo == "ddpg"
                                   else ("c1_flat", "c2_flat", "c1_target_flat", "log_alpha", "_alpha_v")):
        assert torch.equal(getattr(eager, name), getattr(graph, name)), name
    opt = "critic_optimizer" if algo == "ddpg" else "critic1_optim"
    assert getattr(eager, opt).step_count == getattr(graph, opt).step_count > 30


@pytest.mark.parametrize("algo", ["rainbow", "sac", "sac_fused"])
def test_baseline_config_sizes_graphed_equals_eager(algo):
    """BASELINE configs 3 and 4 at their own sizes — Rainbow DQN CartPole-v1 with 8192 envs (2^20-leaf PER tree, n-step
    windows, NoisyNet) and SAC Pendulum-v1 with 4096 envs — for 12 vector steps: the hipGraph-replayed update equals the
    eager one bit for bit and every network / the float64 sum tree is finite.  "sac_fused" is the DEFAULT SAC path (what
    bench.py --algo sac times): the five-launch fused step, eager against its 16-step StepChunk graph, 36 vector steps; the
    fused step against the layer-by-layer path at this size is tests/test_fused_step_gpu.py's (4096, 128, 256) case and its
    acting launch against the oracle test_sac_act_step_vs_oracle's."""
    if algo == "rainbow":
        from gymrl_amd import rainbow_dqn_cartpole as mod
        cls, n_envs = "RainbowDQNTrainer", 8192

        def setup(cfg):
            cfg.num_envs, cfg.memory_capacity = n_envs, 1 << 20
    else:
        from gymrl_amd import sac_pendulum as mod
        cls, n_envs = "SACTrainer", 4096

        def setup(cfg):
            cfg.num_envs, cfg.memory_capacity, cfg.fused_step = n_envs, 1 << 20, algo == "sac_fused"   # "sac": the layer-by-layer update's graph
    outs = []
    for graphs in (False, True):
        if algo == "rainbow":
            mod.NoisyLinear._counter = 0
        outs.append(_run(mod, cls, graphs, 36 if algo == "sac_fused" else 12, setup))
    eager, graph = outs
    if algo == "sac_fused":
        assert graph._fused_ok() and graph._chunk is not None and graph._chunk.graph is not None and getattr(eager, "_chunk", None) is None
        for x, y in zip(eager.memory.ring, graph.memory.ring):
            assert torch.equal(x, y)
        assert torch.equal(eager._alpha_m, graph._alpha_m) and torch.equal(eager.actor_optimizer.v, graph.actor_optimizer.v)
    else:
        assert graph._graph is not None and graph._graph.graph is not None
    if algo == "rainbow":
        assert eager.optimizer.step_count == graph.optimizer.step_count >= 4
        assert torch.equal(eager.flat_params, graph.flat_params) and torch.equal(eager.target_flat, graph.target_flat)
        assert torch.equal(eager.memory.sum_tree.tree, graph.memory.sum_tree.tree)
        assert bool(torch.isfinite(graph.memory.sum_tree.tree).all()) and len(graph.memory) == len(eager.memory) > n_envs
    else:
        assert eager.critic_optimizer.step_count == graph.critic_optimizer.step_count >= 4
        for name in ("actor_flat", "critic_flat", "critic_target_flat", "log_alpha"):
            assert torch.equal(getattr(eager, name), getattr(graph, name)), name
        assert bool(torch.isfinite(graph.actor_flat).all()) and len(graph.memory) == len(eager.memory) > n_envs


@pytest.mark.parametrize("hidden,every", [(256, 0), (256, 3), (64, 0)])
def test_ppo_minibatch_graph_equals_eager(hidden, every):
    """PPOTrainer.update(): the minibatch body (gather, the hand-GEMM step(), norm + Adam with device-side bias corrections)
    replayed as a hipGraph == the eager loop bit for bit — parameters, Adam moments, metrics; also when bench.py's sampling
    timers pull every third minibatch out of the replay into the bracketed eager sequence, and across two iterations with
    the annealed learning rate (it travels through the scalar block, not through the capture)."""
    from gymrl_amd.ppo_lunarlander import Config, KernelTimers, PPOTrainer

    def run(graphs):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.num_minibatches, cfg.seed = 128, 32, 2, 8, 4
        cfg.hidden_dim, cfg.use_graphs, cfg.max_train_steps = hidden, graphs, 128 * 32 * 4
        tr = PPOTrainer(cfg)
        if every:
            tr._timers = KernelTimers(every=every)
        ms = []
        for it in range(2):
            lr = cfg.lr * (1.0 - tr.step_count / cfg.max_train_steps)
            for g in tr.optimizer.param_groups:
                g["lr"] = lr
            ms.append(tr.update(tr.collect_rollout()))
        return tr, ms
    (a, ma), (b, mb_) = run(False), run(True)
    assert b._graph is not None and b._graph["step"].graph is not None and a._graph is None
    assert a.optimizer.step_count == b.optimizer.step_count == 32
    assert torch.equal(a.flat_params, b.flat_params)
    assert torch.equal(a.optimizer.m, b.optimizer.m) and torch.equal(a.optimizer.v, b.optimizer.v)
    assert ma == mb_
    if every:
        ks = b._timers.summary()
        assert ks["gemm_fwd_256_tanh"]["launches"] == 11            # minibatches 0, 3, 6, ... of the 32 of both updates
