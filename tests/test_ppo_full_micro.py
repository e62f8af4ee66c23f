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
