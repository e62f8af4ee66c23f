"""PPO-full gradient accumulation: a minibatch split into equal micro-batches (bounded activation memory at
4096 envs x F0's minibatch ratio) gives the same update as the whole minibatch in one pass."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graphs", [False, True])
def test_micro_batches_equal_whole_minibatch(graphs):
    from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer

    def run(micro):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed, cfg.mhc_dim = 64, 32, 2, 512, 1, 32
        cfg.micro_batch, cfg.use_graphs = micro, graphs
        torch.manual_seed(5)
        tr = PPOTrainer(cfg)
        ms = []
        for _ in range(2):
            tr.collect_experience()
            adv, ret = tr.compute_advantages()
            ms.append(tr.update_model(adv, ret))
        return tr, ms
    (a, ma), (b, mb) = run(0), run(128)
    assert a.optimizer.step_count == b.optimizer.step_count == 2 * 2 * 4
    # same gradient up to the summation order of four partial sums: parameters after 16 Adam steps
    assert float((a.flat_params - b.flat_params).abs().max()) <= 2e-5
    for x, y in zip(ma, mb):
        for k in x:
            assert abs(x[k] - y[k]) <= 1e-4 * max(1.0, abs(x[k])), (k, x[k], y[k])


def test_update_graphs_are_kept_across_updates_and_follow_the_annealed_coefficients():
    """update_model()'s two graphs (forward / loss / backward of a micro-batch; clip + Adam) are captured once and replayed by
    LATER calls too: the annealed entropy coefficient and learning rate reach the replayed kernels through device memory
    (gymrl_ppo_full_cfg.entropy_coef_dev, Adam's bias block).  Three iterations against the eager loop: parameters, Adam
    moments and every reported metric bit for bit; the graph objects of call 2 are call 3's."""
    from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer

    def run(graphs):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed, cfg.mhc_dim = 64, 32, 2, 512, 1, 32
        cfg.micro_batch, cfg.use_graphs, cfg.max_train_steps = 128, graphs, 64 * 32 * 4        # (annealing visibly: 1, 3/4, 1/2)
        torch.manual_seed(5)
        tr = PPOTrainer(cfg)
        ms, kept, coefs = [], [], []
        for _ in range(3):
            tr.collect_experience()
            adv, ret = tr.compute_advantages()
            coefs.append((tr.ent_coef, tr.lr))
            ms.append(tr.update_model(adv, ret))
            kept.append(getattr(tr, "_g_graphs", None))
        return tr, ms, kept, coefs
    (a, ma, _, ca), (b, mb, kept, cb) = run(False), run(True)
    assert ca == cb and len({c[0] for c in cb}) == 3 and len({c[1] for c in cb}) == 3
    assert kept[0] is not None and kept[1][1] is kept[2][1] and kept[1][2] is kept[2][2] and kept[0][1] is kept[1][1]
    assert torch.equal(a.flat_params, b.flat_params)
    assert torch.equal(a.optimizer.m, b.optimizer.m) and torch.equal(a.optimizer.v, b.optimizer.v)
    assert ma == mb
