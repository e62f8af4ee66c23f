"""Run-to-run determinism of the default paths: the same seed gives the same bits twice.  No kernel on these paths adds
floats through atomics in an order the scheduler chooses (partials are written per workgroup and added in a fixed order;
the sum tree's per-node order is the batch order), the large-batch row kernels take their place from a start-order ticket
that only decides WHO computes a slab, never what is computed, and Rainbow's two HIP queues meet at events — so two runs of a
trainer must agree on every parameter, moment, replay row and tree node, also at the batch sizes that use the ticketed
1-D grids (SAC at 1024 rows, Rainbow at 2048)."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _twice(fn):
    a = fn()
    torch.cuda.synchronize()
    b = fn()
    torch.cuda.synchronize()
    return a, b


def test_ppo_two_runs_same_bits():
    from gymrl_amd.ppo_lunarlander import Config, PPOTrainer

    def run():
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.num_minibatches, cfg.seed = 512, 64, 2, 4, 9
        tr = PPOTrainer(cfg)
        ms = [tr.update(tr.collect_rollout()) for _ in range(2)]
        b = tr.buffer
        return dict(p=tr.flat_params.clone(), m=tr.optimizer.m.clone(), v=tr.optimizer.v.clone(), adv=b.advantages.clone(),
                    states=b.states.clone(), actions=b.actions.clone(), values=b.values.clone()), ms
    (a, ma), (b, mb) = _twice(run)
    assert ma == mb
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_ppo_full_two_runs_same_bits():
    from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer

    def run():
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed, cfg.mhc_dim = 128, 32, 2, 512, 3, 32
        torch.manual_seed(7)
        tr = PPOTrainer(cfg)
        ms = []
        for _ in range(3):                       # the third update replays the graphs the second one captured
            tr.collect_experience()
            adv, ret = tr.compute_advantages()
            ms.append(tr.update_model(adv, ret))
        return dict(p=tr.flat_params.clone(), m=tr.optimizer.m.clone(), v=tr.optimizer.v.clone(), adv=adv.clone(),
                    actions=tr.buffer.actions.clone()), ms
    (a, ma), (b, mb) = _twice(run)
    assert ma == mb
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("batch", [128, 1024])
def test_sac_two_runs_same_bits(batch):
    from gymrl_amd import sac_pendulum as mod

    def run():
        cfg = mod.Config()
        cfg.num_envs, cfg.max_episodes, cfg.batch_size, cfg.seed, cfg.memory_capacity = 512, 10**9, batch, 5, 1 << 16
        torch.manual_seed(11)
        tr = mod.SACTrainer(cfg)
        tr.train(max_vector_steps=40)
        torch.cuda.synchronize()
        assert tr._fused_ok()
        out = {n: getattr(tr, n).clone() for n in ("actor_flat", "critic_flat", "critic_target_flat", "log_alpha", "_alpha_m", "_alpha_v")}
        out.update({f"ring{i}": r.clone() for i, r in enumerate(tr.memory.ring)})
        out["steps"] = torch.tensor([tr.critic_optimizer.step_count])
        return out
    a, b = _twice(run)
    assert int(a["steps"]) >= 30
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("batch", [256, 2048])
def test_rainbow_two_runs_same_bits(batch):
    from gymrl_amd import rainbow_dqn_cartpole as mod

    def run():
        mod.NoisyLinear._counter = 0
        cfg = mod.Config()
        cfg.num_envs, cfg.max_episodes, cfg.batch_size, cfg.seed, cfg.memory_capacity = 1024, 10**9, batch, 5, 1 << 16
        torch.manual_seed(11)
        tr = mod.RainbowDQNTrainer(cfg)
        tr.train(max_vector_steps=40)
        torch.cuda.synchronize()
        return dict(p=tr.flat_params.clone(), t=tr.target_flat.clone(), tree=tr.memory.sum_tree.tree.clone(),
                    m=tr.optimizer.m.clone(), steps=torch.tensor([tr.optimizer.step_count]))
    a, b = _twice(run)
    assert int(a["steps"]) >= 20
    for k in a:
        assert torch.equal(a[k], b[k]), k
