#!/usr/bin/env python3
"""Is a graphed off-policy vector step bound by the host (issuing its eager launches) or by the GPU?  Times the
ENQUEUE of K steps (no synchronisation inside) against the same K steps including the final device sync."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import rainbow_dqn_cartpole, sac_pendulum  # noqa: E402
from gymrl_amd.envs import EpisodeTracker  # noqa: E402


def main():
    out = {}
    for mod, cls, N, B in ((rainbow_dqn_cartpole, "RainbowDQNTrainer", 8192, 256), (sac_pendulum, "SACTrainer", 4096, 128)):
        c = mod.Config()
        c.num_envs, c.memory_capacity, c.max_episodes, c.batch_size = N, 1 << 20, 10**9, B
        tr = getattr(mod, cls)(c)
        sys.stdout = open(os.devnull, "w")
        tr.train(max_vector_steps=80)
        sys.stdout = sys.__stdout__
        # one more train() call whose EpisodeTracker never flushes inside the timed window
        orig = EpisodeTracker.advance
        EpisodeTracker.advance = lambda self, sink: None
        tr.env = type(tr.env)(tr.cfg.env_name, tr.cfg.num_envs, device=tr.device, seed=tr.base_seed)
        K = 200
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sys.stdout = open(os.devnull, "w")
        tr.train(max_vector_steps=K)          # ends with tracker.flush (one sync) -> measure enqueue via a hook below
        sys.stdout = sys.__stdout__
        torch.cuda.synchronize()
        t_total = time.perf_counter() - t0
        EpisodeTracker.advance = orig
        out[cls] = dict(ms_per_step_total=round(t_total / K * 1e3, 3), enqueue_ms_per_step=round(_ENQ.pop() / K * 1e3, 3))
    print(json.dumps(out, indent=1))


_ENQ = []
_flush = EpisodeTracker.flush


def _timed_flush(self, sink):
    _ENQ.append(time.perf_counter() - _T0[0])
    return _flush(self, sink)


_T0 = [0.0]
_reset = None


if __name__ == "__main__":
    # enqueue time = from the start of train() to the moment the loop reaches its final tracker.flush()
    EpisodeTracker.flush = _timed_flush
    import gymrl_amd.envs as _envs
    _orig_reset = _envs.VecEnv.reset

    def _r(self, *a, **k):
        torch.cuda.synchronize()
        _T0[0] = time.perf_counter()
        return _orig_reset(self, *a, **k)

    _envs.VecEnv.reset = _r
    main()
