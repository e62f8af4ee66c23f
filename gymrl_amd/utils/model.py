"""Checkpoint mixin of the legacy agent contract — the reference's `ModelLoader` (utils/model.py:330-366).

An agent that derives from it gets `save_model()` / `load_model()` writing ONE dict to `cfg.save_path`:
`<attr>_state_dict` for every attribute that has a `state_dict()` (networks, `FusedAdam(module=net)` in
torch.optim.Adam's layout, `Normalization` / `RewardScaling` statistics) and the plain attributes (`learn_step`,
...) under their own names; `cfg`, `memory` and `state_buffer` are skipped, as in the reference.  utils/runner.py
calls `save_model()` every `cfg.save_freq` finished episodes and `load_model()` when `cfg.load_model` is set.
"""
import os
import pickle

import torch

_SKIP = ("state_buffer", "cfg", "memory")


def _to_cpu(x):
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: _to_cpu(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cpu(v) for v in x)
    return x


class ModelLoader:
    def __init__(self, cfg):
        cfg.save_path = f'./checkpoints/{cfg.algo_name}_{cfg.env_name.replace("/", "-")}.pth'
        self.cfg = cfg
        os.makedirs(os.path.dirname(cfg.save_path), exist_ok=True)

    def save_model(self):
        state = {}
        for key, value in self.__dict__.items():
            if key in _SKIP or key.startswith("_"):
                continue
            if hasattr(value, "state_dict"):
                state[f"{key}_state_dict"] = _to_cpu(value.state_dict())
            elif torch.is_tensor(value):
                state[key] = value.detach().cpu()
            elif isinstance(value, (int, float, str, bool, list, tuple, dict, type(None))):
                state[key] = value
            else:       # the reference keeps every non-excluded attribute (numpy arrays, deques of returns, ...)
                try:
                    pickle.dumps(value)
                    state[key] = value
                except Exception:                     # locks, open files, compiled graphs: say so instead of dropping silently
                    print(f"[save_model] skipping attribute {key!r} ({type(value).__name__}: not picklable)")
        torch.save(state, self.cfg.save_path)
        return state

    def load_model(self):
        ck = torch.load(self.cfg.save_path, map_location="cpu", weights_only=False)
        for key, value in ck.items():
            if key in _SKIP:
                continue
            if key.endswith("_state_dict"):
                attr = key[:-len("_state_dict")]
                if hasattr(self, attr):
                    getattr(self, attr).load_state_dict(value)
                else:       # e.g. state_norm / reward_scaler, which runner.train() attaches after load_model()
                    self.__dict__.setdefault("_pending_state", {})[attr] = value
            elif torch.is_tensor(value) and torch.is_tensor(getattr(self, key, None)):
                getattr(self, key).copy_(value.to(getattr(self, key).device))     # device buffers keep their storage
            else:
                setattr(self, key, value)
        return ck
