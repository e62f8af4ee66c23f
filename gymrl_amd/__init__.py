"""gymrl_amd — MI355X-native vectorised-rollout + PPO/DQN/SAC update engine that
sits behind the Config / *Trainer surface of Starlight0798/gymRL's algorithms/*.py.

The compute path is libgymrl_hip.so (hand-written gfx950 HIP kernels behind the
C-ABI of include/gymrl.h).  There is no CPU fallback: ops raise if the library or
an MI355X is missing.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
