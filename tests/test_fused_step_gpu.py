"""The fused SAC vector step (csrc/offpolicy_step.hip: acting + env + replay row in one launch, update() in four) against the
layer-by-layer path it replaces (gymrl_lin_* launches + the stand-alone loss / optimiser / replay / env kernels, which
tests/test_trainers_gpu.py pins against the reference's own update() and train()): same noise, same index draws ->
every parameter, Adam moment, the target network, the float64 temperature and the replay ring equal BIT FOR BIT."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(fused, steps, N, B, hidden, seed=5, images=True):
    from gymrl_amd.sac_pendulum import Config, SACTrainer
    cfg = Config()
    cfg.num_envs, cfg.batch_size, cfg.hidden_dim, cfg.seed = N, B, hidden, seed
    cfg.max_episodes, cfg.memory_capacity, cfg.use_graphs, cfg.fused_step, cfg.fused_images = 10 ** 9, (1 << 20 if N >= 4096 else max(4096, 4 * B)), False, fused, images
    tr = SACTrainer(cfg)
    assert tr._fused_ok() == fused
    g = torch.Generator(device="cuda").manual_seed(7)
    A = tr.env.act_dim
    tr._parity_eps = iter([torch.randn(N, A, generator=g, device="cuda") for _ in range(steps)])
    # explicit N(0,1) draws for both samples of every update; indices None: each path draws its own (the eager path through
    # gymrl_uniform_indices, the fused path inside P1) from the same (seed, counter, size)
    tr._parity_updates = iter([(None, torch.randn(B, A, generator=g, device="cuda"), torch.randn(B, A, generator=g, device="cuda"))
                               for _ in range(steps)])
    tr.train(max_vector_steps=steps)
    torch.cuda.synchronize()
    return tr


# (hidden 256: the instances built for that width; 36: no weight images, no 16-column alignment; B = 100: a partial last slab;
#  B = 1024 / 2100: the large-batch form — slab-adjacent 1-D grids, the weight gradients' 256-row slices and the two-level loss
#  sums of the layer-by-layer kernels beyond 512 / 256 rows (SURVEY 8(d)'s B = 4096 line runs this code);
#  4096 / 128 / 256 is BASELINE config 4 itself: sac_act_kernel<256> on 256 workgroups, ring of 2^20 rows)
@pytest.mark.parametrize("N,B,hidden,steps", [(64, 128, 256, 40), (20, 24, 32, 30), (48, 256, 64, 24), (33, 100, 36, 20), (17, 250, 256, 30),
                                              (4096, 128, 256, 12), (512, 1024, 256, 7), (1100, 2100, 64, 5)])
def test_sac_fused_step_equals_layer_by_layer(N, B, hidden, steps):
    a, b = _run(False, steps, N, B, hidden), _run(True, steps, N, B, hidden)
    assert a.critic_optimizer.step_count == b.critic_optimizer.step_count > (10 if B <= 256 else 2)
    assert (a.memory.cursor, a.memory.size, a.memory.draws) == (b.memory.cursor, b.memory.size, b.memory.draws)
    for x, y in zip(a.memory.ring, b.memory.ring):
        assert torch.equal(x, y)                                    # acting: same actions, same physics, same rows
    for name in ("actor_flat", "critic_flat", "critic_target_flat", "log_alpha", "_alpha_m", "_alpha_v"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    for opt in ("actor_optimizer", "critic_optimizer"):
        assert torch.equal(getattr(a, opt).m, getattr(b, opt).m) and torch.equal(getattr(a, opt).v, getattr(b, opt).v), opt
    assert torch.equal(a._sums[:3], b._sums[:3]) and torch.equal(a._alpha_loss, b._alpha_loss)
    assert list(a.episode_rewards) == list(b.episode_rewards)
    if hidden % 16 == 0:
        # the weight images (MFMA operands of the hidden x hidden layers, kept in step by the weight-gradient kernel) change
        # where a value is read from, not the value: without them the same bits again
        c = _run(True, steps, N, B, hidden, images=False)
        assert b._fused[4] is not None and c._fused[4] is None
        for name in ("actor_flat", "critic_flat", "critic_target_flat", "log_alpha"):
            assert torch.equal(getattr(b, name), getattr(c, name)), name
        # and they do hold the parameters: rebuilding them from scratch changes nothing
        before = b._fused[4].clone()
        from gymrl_amd import ops
        ops.sac_pack_images(b._fused[1])
        assert torch.equal(before, b._fused[4])


def test_sac_fused_update_matches_reference_update():
    """The reference's own SACTrainer.update() (tests/golden/sac.npz: one update from fixed weights, batch, N(0,1) draws)
    reproduced by the FUSED update: losses, the float64 temperature, actor / critic / target parameters."""
    from conftest import load_golden
    from gymrl_amd.sac_pendulum import Config, SACTrainer
    g = load_golden("sac")
    cfg = Config()
    cfg.hidden_dim, cfg.batch_size = 32, 24
    cfg.gamma, cfg.tau = float(g["u_gamma"]), float(g["u_tau"])
    tr = SACTrainer(cfg)
    assert tr._fused_update_ok()
    for name, net in (("actor", tr.actor), ("critic", tr.critic), ("critic_target", tr.critic_target)):
        pre = f"u0_{name}_"
        net.load_state_dict({k[len(pre):]: torch.from_numpy(np.array(g[k])) for k in g.files
                             if k.startswith(pre) and not k[len(pre):].startswith("target_")})
    dev = tr.device
    tr.memory.push(torch.from_numpy(g["u_states"]).to(dev), torch.from_numpy(g["u_actions"]).to(dev),
                   torch.from_numpy(g["u_rewards"]).to(dev), torch.from_numpy(g["u_next_states"]).to(dev),
                   torch.from_numpy(g["u_dones"]).to(dev))
    al, cl, aal = tr.update(indices=torch.from_numpy(g["u_order"]).to(dev), eps_next=torch.from_numpy(g["u_eps_next"]).to(dev),
                            eps_cur=torch.from_numpy(g["u_eps_cur"]).to(dev))
    ref = g["u_losses"]
    for got, want in ((al, ref[0]), (cl, ref[1]), (aal, ref[2])):
        assert abs(got - want) <= 2e-5 * max(1, abs(want))
    assert abs(tr.log_alpha.item() - float(g["u_log_alpha1"])) <= 1e-9
    for name, net in (("actor", tr.actor), ("critic", tr.critic), ("critic_target", tr.critic_target)):
        err = max(float(np.max(np.abs(v.detach().cpu().numpy() - g[f"u1_{name}_" + k]))) for k, v in net.state_dict().items())
        assert err <= 5e-6, (name, err)


def test_sac_fused_step_is_one_launch_and_equals_its_five():
    """What the fusion is for: a vector step of the chunked graph is ONE kernel node (gymrl_sac_step: acting | rows | tiles |
    rows | tiles as block ranges of one grid, handing over through counters in the workspace) — and what it computes is what
    the five launches it replaces compute, bit for bit (48 launches in a row: the counters come back to zero each time)."""
    from gymrl_amd.sac_pendulum import Config, SACTrainer
    outs = []
    for one, (N, B, hidden) in [(o, shp) for shp in ((256, 128, 256), (4100, 256, 256), (40, 24, 32)) for o in (True, False)]:
        cfg = Config()
        cfg.num_envs, cfg.batch_size, cfg.hidden_dim, cfg.seed, cfg.max_episodes = N, B, hidden, 1, 10 ** 9
        cfg.memory_capacity, cfg.one_launch_step = max(1 << 14, 2 * N), one
        torch.manual_seed(3)
        tr = SACTrainer(cfg)
        tr.train(max_vector_steps=64)
        torch.cuda.synchronize()
        assert tr._fused_ok() and tr._chunk is not None and tr._chunk.graph is not None
        assert all(torch.isfinite(getattr(tr, n)).all() for n in ("actor_flat", "critic_flat", "log_alpha"))
        assert tr.critic_optimizer.step_count >= 48
        outs.append(tr)
        if not one:
            a, b = outs[-2], outs[-1]
            for name in ("actor_flat", "critic_flat", "critic_target_flat", "log_alpha", "_alpha_m", "_alpha_v", "_sums"):
                assert torch.equal(getattr(a, name), getattr(b, name)), (N, B, hidden, name)
            for x, y in zip(a.memory.ring, b.memory.ring):
                assert torch.equal(x, y)
            assert list(a.episode_rewards) == list(b.episode_rewards)


def _run_rainbow(fused, steps, N, B, hidden, graphs=False, chunk=0, images=True):
    from gymrl_amd import rainbow_dqn_cartpole as rb
    rb.NoisyLinear._counter = 0
    cfg = rb.Config()
    cfg.fused_images = images
    cfg.num_envs, cfg.batch_size, cfg.hidden_dim, cfg.seed = N, B, hidden, 5
    cfg.max_episodes, cfg.memory_capacity, cfg.use_graphs, cfg.chunk_steps, cfg.fused_step = 10 ** 9, max(1 << 12, 2 * N, 2 * B), graphs, chunk, fused
    torch.manual_seed(11)
    tr = rb.RainbowDQNTrainer(cfg)
    assert tr._fused_act_ok() == fused
    tr.train(max_vector_steps=steps)
    torch.cuda.synchronize()
    return tr


@pytest.mark.parametrize("N,B,hidden,steps", [(64, 256, 256, 70), (20, 24, 32, 40), (128, 128, 64, 50), (8192, 256, 256, 12), (4100, 64, 32, 9),
                                              (1024, 2048, 256, 10), (8192, 8192, 256, 11)])
def test_rainbow_fused_step_equals_layer_by_layer(N, B, hidden, steps):
    """Rainbow: acting + env + n-step push as one launch and the update's Linear / loss / backward launches as two, against
    the layer-by-layer path (tests/test_trainers_gpu.py pins that one to the reference): networks, Adam moments, the float64
    sum tree, the replay ring and the n-step windows bit for bit (the ring wraps: 4096 rows)."""
    a, b = _run_rainbow(False, steps, N, B, hidden), _run_rainbow(True, steps, N, B, hidden)
    assert a.optimizer.step_count == b.optimizer.step_count > 4
    assert (a.total_steps, a.memory.count, a.memory.pushes, a.memory.draws) == (b.total_steps, b.memory.count, b.memory.pushes, b.memory.draws)
    for x, y in zip(a.memory.ring + a.memory.win, b.memory.ring + b.memory.win):
        assert torch.equal(x, y)
    assert torch.equal(a.memory.sum_tree.tree, b.memory.sum_tree.tree)
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.target_flat, b.target_flat)
    assert torch.equal(a.optimizer.m, b.optimizer.m) and torch.equal(a.optimizer.v, b.optimizer.v)
    assert torch.equal(a._loss, b._loss)
    assert a.optimizer.param_groups[0]["lr"] == b.optimizer.param_groups[0]["lr"]
    for name in ("advantage", "value"):
        assert torch.equal(getattr(a.policy_net, name).weight_epsilon, getattr(b.policy_net, name).weight_epsilon)
    assert list(a.episode_rewards) == list(b.episode_rewards)


def test_rainbow_fused_chunked_equals_fused_eager():
    a = _run_rainbow(True, 70, 64, 128, 256)
    b = _run_rainbow(True, 70, 64, 128, 256, graphs=True, chunk=16)
    assert b._chunk.graph is not None
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.target_flat, b.target_flat)
    assert torch.equal(a.memory.sum_tree.tree, b.memory.sum_tree.tree)
    for x, y in zip(a.memory.ring, b.memory.ring):
        assert torch.equal(x, y)
    assert list(a.episode_rewards) == list(b.episode_rewards)


def test_rainbow_weight_images():
    """gymrl_noisy_combine_images: the images are lin_device.hpp's img_fwd_index / img_bwd_index permutations of the weight,
    the stacked heads are gymrl_noisy_combine's; a chunked run streaming fc2 from them equals one reading it in place."""
    from gymrl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4)
    for H in (256, 48):
        steps = H // 16
        W = torch.randn(H, H, device=dev, generator=g)
        layer = dict(w_mu=torch.randn(3, H, device=dev, generator=g), w_sigma=torch.rand(3, H, device=dev, generator=g),
                     b_mu=torch.randn(3, device=dev, generator=g), b_sigma=torch.rand(3, device=dev, generator=g), draw=True, seed=7, counter=5)
        img = torch.full((2, H * H), float("nan"), device=dev)
        Wh, bh = ops.noisy_combine([layer], training=True, images=[(W, img[0], img[1])])
        Wr, br = ops.noisy_combine([layer], training=True)
        assert torch.equal(Wh, Wr) and torch.equal(bh, br)
        n, k = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
        fwd = (((n >> 4) * steps + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3)
        bwd = (((k >> 4) * steps + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (k & 15)) * 4 + (n & 3)
        Wn = W.cpu().numpy()
        exp_f, exp_b = np.empty(H * H, np.float32), np.empty(H * H, np.float32)
        exp_f[fwd.ravel()] = Wn.ravel()
        exp_b[bwd.ravel()] = Wn.ravel()
        assert np.array_equal(img[0].cpu().numpy(), exp_f) and np.array_equal(img[1].cpu().numpy(), exp_b)
        only_b = torch.zeros(H * H, device=dev)
        ops.noisy_combine([layer], training=True, images=[(W, None, only_b)])
        assert torch.equal(only_b, img[1])
    a = _run_rainbow(True, 40, 64, 128, 256, graphs=True, chunk=16, images=False)
    b = _run_rainbow(True, 40, 64, 128, 256, graphs=True, chunk=16)
    assert a._fused_state()["img"] is None and b._fused_state()["img"] is not None
    assert torch.equal(a.flat_params, b.flat_params) and torch.equal(a.target_flat, b.target_flat)
    assert torch.equal(a.memory.sum_tree.tree, b.memory.sum_tree.tree)
    assert list(a.episode_rewards) == list(b.episode_rewards)


@pytest.mark.parametrize("N,hidden,steps", [(4096, 256, 24), (50, 64, 230)])
def test_sac_act_step_vs_oracle(N, hidden, steps):
    """gymrl_sac_act_step (Actor forward + reparameterised draw + Pendulum step + replay row, ONE launch; sac_pendulum.py:278-283)
    against the oracle AT CONFIG 4's SIZE (4096 envs, hidden 256: sac_act_kernel<256> on 256 workgroups), each side acting on
    ITS OWN observations as tests/test_hip_parity.py::test_classic_env_vs_oracle does: the oracle's forward is orc_linear_act's
    fmaf chain in the MFMA order (relu trunk, mean / clamped log_std heads), its draw orc_sac_sample_fwd, its env orc_env_step.
    Actions, rewards, done flags, next observations and the ring rows (state, action, reward, terminal observation, done) are
    compared with array_equal at every step; the small case runs across the 200-step time limit (every env truncates and resets
    once) and wraps its ring."""
    from gymrl_amd import ops
    from gymrl_amd.sac_pendulum import Config, SACTrainer
    from oracle import oracle
    cfg = Config()
    cfg.num_envs, cfg.hidden_dim, cfg.seed, cfg.batch_size = N, hidden, 9, 128
    cfg.memory_capacity = 1 << 20 if N == 4096 else 4 * N           # (config 4's ring; the small one wraps 57 times)
    tr = SACTrainer(cfg)
    assert tr._fused_ok()
    env, m, dev = tr.env, tr.memory, tr.device
    D, A, cap = env.obs_dim, m.ring[1].shape[1], m.capacity
    W = {k: v.detach().cpu().numpy() for k, v in tr.actor.state_dict().items()}
    ref_env = oracle.Env(oracle.PENDULUM, N, seed=env.seed, env_id0=env.env_id0)
    o_ref = ref_env.reset()
    obs, nxt = torch.empty(N, D, device=dev), torch.empty(N, D, device=dev)
    rew, ep_ret = torch.empty(N, device=dev), torch.zeros(N, device=dev)
    done = torch.empty(N, dtype=torch.uint8, device=dev)
    env.reset(obs)
    assert np.array_equal(obs.cpu().numpy(), o_ref)
    g = torch.Generator(device="cuda").manual_seed(21)
    n_done = 0
    for s in range(steps):
        eps = torch.randn(N, A, generator=g, device=dev)
        cursor = m.cursor
        ops.sac_act_step(tr._fused_args()[0], env, obs, nxt, cursor=cursor, eps=eps, rew_out=rew, done_out=done, ep_ret_out=ep_ret,
                         ep_stats=env.ep_stats)
        m.advance(N)
        h = oracle.linear_act(oracle.linear_act(o_ref, W["fc1.weight"], W["fc1.bias"], 2), W["fc2.weight"], W["fc2.bias"], 2)
        mean = oracle.linear_act(h, W["mean.weight"], W["mean.bias"], 0)
        log_std = np.minimum(np.maximum(oracle.linear_act(h, W["log_std.weight"], W["log_std.bias"], 0), np.float32(cfg.log_std_min)),
                             np.float32(cfg.log_std_max))
        act_ref = oracle.sac_sample_fwd(mean, log_std, eps.cpu().numpy(), float(tr.action_bound))[0]
        r = ref_env.step(act_ref)
        rows = (torch.arange(N, device=dev) + cursor) % cap
        ring = [t[rows].cpu().numpy() for t in m.ring]
        assert np.array_equal(ring[1].view(np.float32).reshape(N, A), act_ref), s       # the action words of the ring
        assert np.array_equal(ring[0], o_ref) and np.array_equal(ring[2].reshape(-1), r["rew"]), s
        assert np.array_equal(ring[3], r["term_obs"]) and np.array_equal(ring[4].reshape(-1) != 0, r["done"] != 0), s
        assert np.array_equal(rew.cpu().numpy(), r["rew"]) and np.array_equal(done.cpu().numpy() != 0, r["done"] != 0), s
        assert np.array_equal(nxt.cpu().numpy(), r["obs"]), s
        n_done += int((r["done"] != 0).sum())
        o_ref = r["obs"]
        obs, nxt = nxt, obs
    assert n_done >= (N if steps >= 200 else 0)
