"""utils/normalization.py:4-52 on the device: RunningMeanStd / Normalization / RewardScaling.

The reference feeds ONE observation (or reward) per call; here a call takes the N rows of a
vector step and consumes them in env order 0..N-1 through gymrl_running_norm /
gymrl_reward_scaling — bit-identical to N successive reference calls (float32 mean, float64
S and std, the n == 1 `std = x` quirk, population std).
"""
import torch

from .. import ops


class RunningMeanStd:
    def __init__(self, shape, device="cuda"):
        self.D = int(shape) if not isinstance(shape, (tuple, list)) else int(shape[0])
        # (n, unused, mean[D], S[D], std[D]) — the layout include/gymrl.h documents
        self.stats = torch.zeros(2 + 3 * self.D, dtype=torch.float64, device=device)

    @property
    def n(self):
        return int(self.stats[0].item())

    @property
    def mean(self):
        return self.stats[2:2 + self.D]

    @property
    def S(self):
        return self.stats[2 + self.D:2 + 2 * self.D]

    @property
    def std(self):
        return self.stats[2 + 2 * self.D:2 + 3 * self.D]

    def update(self, x):
        ops.running_norm(x.view(-1, self.D), self.stats, update=True)

    def state_dict(self):
        return {"stats": self.stats.detach().cpu().clone()}

    def load_state_dict(self, sd):
        self.stats.copy_(sd["stats"].to(self.stats.device, torch.float64))


class Normalization:
    def __init__(self, shape, device="cuda"):
        self.running_ms = RunningMeanStd(shape, device)

    def __call__(self, x, update=True):
        """x f32[N, D] (or [D]) -> (x - mean) / (std + 1e-8); update=False for evaluation (:29-35)."""
        flat = x.reshape(-1, self.running_ms.D).contiguous()
        return ops.running_norm(flat, self.running_ms.stats, update=update).view_as(x)

    def state_dict(self):
        """Checkpoints carry the running statistics (the reference pickles the object itself)."""
        return self.running_ms.state_dict()

    def load_state_dict(self, sd):
        self.running_ms.load_state_dict(sd)


class RewardScaling:
    def __init__(self, shape, gamma, num_envs=1, device="cuda"):
        self.shape, self.gamma = shape, float(gamma)
        self.running_ms = RunningMeanStd(1, device)
        self.R = torch.zeros(num_envs, dtype=torch.float64, device=device)

    def __call__(self, x, done=None):
        """x f32[N] rewards -> x / (std(R) + 1e-8) with R = gamma*R + x per env (:44-49).
        `done` u8[N]: zero R after use where an episode ended (reset() at the next episode start)."""
        return ops.reward_scaling(x.contiguous(), done, self.gamma, self.R, self.running_ms.stats)

    def reset(self):
        self.R.zero_()

    def state_dict(self):
        return {"stats": self.running_ms.stats.detach().cpu().clone(), "R": self.R.detach().cpu().clone()}

    def load_state_dict(self, sd):
        self.running_ms.load_state_dict(sd)
        R = sd.get("R")
        if R is not None and tuple(R.shape) == tuple(self.R.shape):
            self.R.copy_(R.to(self.R.device, torch.float64))
        else:               # a different num_envs: the running statistics carry over, the per-env returns restart (= reset())
            self.R.zero_()
