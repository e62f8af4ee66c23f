"""The reference's SCALAR surface on top of the batched engine.

A caller written against the reference hands one numpy observation to `select_action` and gets a python `int`
(dqn_cartpole.py:124-133, rainbow_dqn_cartpole.py:293-309) or an `np.ndarray[act_dim]` (sac_pendulum.py:202-211) back,
pushes python scalars into the replay buffer (dqn_cartpole.py:183), and reads python floats from `collect_rollout` /
numpy arrays from `compute_gae` (ppo_lunarlander.py:179-231).  The engine's own loops pass device tensors and get device
tensors: the return type follows the argument type, so nothing on the hot path changes and no caller has to choose.
Host arguments cost one small H2D copy and one D2H sync per call — the price of the scalar surface, as in the reference.
"""
import numpy as np
import torch

ONE, MANY = "one", "many"


def host_kind(x):
    """None for a device tensor (engine path); ONE for a single host observation [D] / scalar; MANY for a host batch."""
    if torch.is_tensor(x) and x.is_cuda:
        return None
    return ONE if np.ndim(x.cpu().numpy() if torch.is_tensor(x) else x) <= 1 else MANY


def obs_batch(state, device):
    """-> (f32 [N, D] on `device`, kind)."""
    kind = host_kind(state)
    if kind is None:
        return state, None
    a = np.asarray(state.cpu().numpy() if torch.is_tensor(state) else state, dtype=np.float32)
    t = torch.from_numpy(np.ascontiguousarray(a.reshape(1, -1) if kind == ONE else a)).to(device)
    return t, kind


def discrete_out(action, kind):
    """i32 [N] device tensor -> itself | python int | np.int64 [N] (the reference's `.argmax().item()`)."""
    if kind is None:
        return action
    a = action.cpu().numpy()
    return int(a[0]) if kind == ONE else a.astype(np.int64)


def continuous_out(action, kind):
    """f32 [N, A] device tensor -> itself | np.ndarray [A] | np.ndarray [N, A] (`.cpu().numpy().flatten()`)."""
    if kind is None:
        return action
    a = action.cpu().numpy()
    return a[0].copy() if kind == ONE else a


def rows(x, n, dtype, device, width=None):
    """One replay-row field, host or device, scalar or array -> a device tensor [n] / [n, width] of `dtype`."""
    if torch.is_tensor(x) and x.is_cuda:
        return x
    a = np.asarray(x.cpu().numpy() if torch.is_tensor(x) else x)
    a = a.reshape(n) if width is None else a.reshape(n, width)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
