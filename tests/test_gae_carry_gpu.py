"""gymrl_gae variant 3: the persistent LunarLander rollout runs the blocked scan's carry pass for its own envs at its tail
(gymrl_device.hpp gae_carry_scan restates gae.hip's gae_blk_carry_kernel per lane), so compute_gae is the apply launch
alone.  Same operations in the same order: advantages, returns and moments must equal variant 2's bit for bit — also when
the rollout has more than one round of 128 chunks, a ragged last chunk, or is cut into several launches."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,N,chunk", [(64, 64, 0), (100, 48, 0), (2100, 32, 0), (2048, 256, 512), (33, 20, 16)])
def test_variant3_equals_variant2(T, N, chunk):
    from gymrl_amd.ppo_lunarlander import Config, PPOTrainer

    def run(carry):
        cfg = Config()
        cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.num_minibatches, cfg.seed = N, T, 1, 2, 21
        cfg.rollout_chunk, cfg.gae_carry_in_rollout = chunk, carry
        tr = PPOTrainer(cfg)
        tr.collect_rollout()
        adv, ret = tr.compute_gae()
        return tr, adv.clone(), ret.clone(), tr._moments.clone(), tr.buffer.values.clone()
    a, adv3, ret3, mom3, val3 = run(True)
    b, adv2, ret2, mom2, val2 = run(False)
    assert a._last_gae_variant == 3 and b._last_gae_variant == 2
    assert torch.equal(val3, val2)
    assert torch.equal(adv3, adv2) and torch.equal(ret3, ret2) and torch.equal(mom3, mom2)
    # ... and both equal the self-contained blocked scan over the same slab (variant 1)
    adv1, ret1 = a.compute_gae()
    assert a._last_gae_variant == 1 and torch.equal(adv1, adv3) and torch.equal(ret1, ret3)
