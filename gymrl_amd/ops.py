"""Thin torch-tensor front-end over the C-ABI (include/gymrl.h).

PyTorch is plumbing here: it owns device memory and the stream; every op below
passes raw device pointers + the current HIP stream to libgymrl_hip.so.  No op
has a CPU path — tensors must live on an MI355X.
"""
import ctypes as C

import torch

from ._lib import PPOCfg, PPOFullCfg, check, lib

_vp = C.c_void_p


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return _vp(None)
        raise ValueError("tensor required")
    if not t.is_cuda:
        raise RuntimeError("gymrl_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return _vp(t.data_ptr())


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def device_ok():
    return bool(lib().gymrl_device_ok())


# ------------------------------------------------------------------ GAE -----
def gae_workspace(T, N, device):
    nbytes = lib().gymrl_gae_workspace_bytes(C.c_int(T), C.c_int(N))
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def reduce_workspace(device):
    return torch.empty(int(lib().gymrl_reduce_workspace_bytes()), dtype=torch.uint8, device=device)


def gae(rew, val, done, next_val, gamma, lam, adv_out=None, ret_out=None, moments_out=None, variant=1,
        workspace=None):
    """G1 (ppo_lunarlander.py:179-196) over a [T][N] slab.  Returns (adv, ret)."""
    T, N = rew.shape
    adv_out = torch.empty_like(rew) if adv_out is None else adv_out
    ret_out = torch.empty_like(rew) if ret_out is None else ret_out
    if workspace is None and (variant == 1 or moments_out is not None):
        workspace = gae_workspace(T, N, rew.device)
    check(lib().gymrl_gae(_ptr(rew, torch.float32), _ptr(val, torch.float32), _ptr(done, torch.uint8),
                          _ptr(next_val, torch.float32), C.c_int(T), C.c_int(N), C.c_double(gamma),
                          C.c_double(lam), _ptr(adv_out, torch.float32), _ptr(ret_out, torch.float32),
                          _ptr(moments_out, torch.float64, True), C.c_int(variant),
                          _ptr(workspace, None, True), _stream()), "gymrl_gae")
    return adv_out, ret_out


def gae_dw(rew, val, next_val, done, dw, gamma, lam, moments_out=None, workspace=None):
    """G2 (utils/buffer.py:21-35).  Returns (adv, v_target)."""
    T, N = rew.shape
    adv, vt = torch.empty_like(rew), torch.empty_like(rew)
    if workspace is None and moments_out is not None:
        workspace = gae_workspace(T, N, rew.device)
    check(lib().gymrl_gae_dw(_ptr(rew, torch.float32), _ptr(val, torch.float32), _ptr(next_val, torch.float32),
                             _ptr(done, torch.uint8), _ptr(dw, torch.uint8), C.c_int(T), C.c_int(N),
                             C.c_double(gamma), C.c_double(lam), _ptr(adv), _ptr(vt),
                             _ptr(moments_out, torch.float64, True), _ptr(workspace, None, True), _stream()),
          "gymrl_gae_dw")
    return adv, vt


def gae_decoupled(rew, val, done, next_val, gamma, lam_actor, lam_critic):
    """G3 (ppo_full_lunarlander.py:507-535).  Returns (adv_actor, returns)."""
    T, N = rew.shape
    adv, ret = torch.empty_like(rew), torch.empty_like(rew)
    check(lib().gymrl_gae_decoupled(_ptr(rew, torch.float32), _ptr(val, torch.float32), _ptr(done, torch.uint8),
                                    _ptr(next_val, torch.float32), C.c_int(T), C.c_int(N), C.c_double(gamma),
                                    C.c_double(lam_actor), C.c_double(lam_critic), _ptr(adv), _ptr(ret),
                                    _stream()), "gymrl_gae_decoupled")
    return adv, ret


def moments(x, out=None, workspace=None):
    out = torch.empty(3, dtype=torch.float64, device=x.device) if out is None else out
    workspace = reduce_workspace(x.device) if workspace is None else workspace
    check(lib().gymrl_moments(_ptr(x, torch.float32), C.c_int64(x.numel()), _ptr(out, torch.float64),
                              _ptr(workspace), _stream()), "gymrl_moments")
    return out


def normalize_(x, mom, ddof=0, eps=1e-8):
    check(lib().gymrl_normalize(_ptr(x, torch.float32), C.c_int64(x.numel()), _ptr(mom, torch.float64),
                                C.c_int(ddof), C.c_double(eps), _stream()), "gymrl_normalize")
    return x


# ------------------------------------------------------------ categorical ---
def categorical_sample(logits, value=None, noise_exp=None, seed=0, counter=0, env_id0=0, deterministic=False,
                       act_out=None, logp_out=None, ent_out=None, value_out=None):
    """P2 (ppo_lunarlander.py:92-104).  Returns (action i32[N], logp, entropy, value_out)."""
    n, A = logits.shape
    dev = logits.device
    act_out = torch.empty(n, dtype=torch.int32, device=dev) if act_out is None else act_out
    logp_out = torch.empty(n, dtype=torch.float32, device=dev) if logp_out is None else logp_out
    ent_out = torch.empty(n, dtype=torch.float32, device=dev) if ent_out is None else ent_out
    if value is not None and value_out is None:
        value_out = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib().gymrl_categorical_sample(_ptr(logits, torch.float32), _ptr(value, torch.float32, True),
                                         _ptr(noise_exp, torch.float32, True), C.c_uint64(seed),
                                         C.c_uint64(counter), C.c_int64(env_id0), C.c_int(n), C.c_int(A),
                                         C.c_int(int(deterministic)), _ptr(act_out, torch.int32), _ptr(logp_out),
                                         _ptr(ent_out, None, True), _ptr(value_out, None, True), _stream()),
          "gymrl_categorical_sample")
    return act_out, logp_out, ent_out, value_out


# --------------------------------------------------------------- PPO loss ---
_ws_cache = {}


def _reduce_ws(device):
    key = (device.type, device.index)
    if key not in _ws_cache:
        _ws_cache[key] = reduce_workspace(device)
    return _ws_cache[key]


def ppo_loss_fwd_bwd(logits, value, act, logp_old, adv, ret, cfg, idx=None, adv_moments=None, dlogits_out=None,
                     dvalue_out=None, metrics_sum=None, workspace=None):
    """L1+L2 (ppo_lunarlander.py:278-322).  cfg = (clip_eps, dual_clip, value_coef, entropy_coef)."""
    B, A = logits.shape
    dlogits_out = torch.empty_like(logits) if dlogits_out is None else dlogits_out
    dvalue_out = torch.empty(B, dtype=torch.float32, device=logits.device) if dvalue_out is None else dvalue_out
    c = PPOCfg(*cfg)
    if metrics_sum is not None and workspace is None:
        workspace = _reduce_ws(logits.device)
    check(lib().gymrl_ppo_loss_fwd_bwd(_ptr(logits, torch.float32), _ptr(value, torch.float32),
                                       _ptr(idx, torch.int32, True), _ptr(act, torch.int32),
                                       _ptr(logp_old, torch.float32), _ptr(adv, torch.float32),
                                       _ptr(ret, torch.float32), _ptr(adv_moments, torch.float64, True),
                                       C.c_int(B), C.c_int(A), C.byref(c), _ptr(dlogits_out), _ptr(dvalue_out),
                                       _ptr(metrics_sum, torch.float64, True), _ptr(workspace, None, True),
                                       _stream()),
          "gymrl_ppo_loss_fwd_bwd")
    return dlogits_out, dvalue_out


def ppo_full_loss_fwd_bwd(logits, value, act, logp_old, ent_old, adv, ret, cfg, idx=None, dlogits_out=None,
                          dvalue_out=None, metrics_sum=None, workspace=None):
    """L3 (ppo_full_lunarlander.py:575-652)."""
    B, A = logits.shape
    dlogits_out = torch.empty_like(logits) if dlogits_out is None else dlogits_out
    dvalue_out = torch.empty(B, dtype=torch.float32, device=logits.device) if dvalue_out is None else dvalue_out
    c = PPOFullCfg(*cfg)
    if metrics_sum is not None and workspace is None:
        workspace = _reduce_ws(logits.device)
    check(lib().gymrl_ppo_full_loss_fwd_bwd(_ptr(logits, torch.float32), _ptr(value, torch.float32),
                                            _ptr(idx, torch.int32, True), _ptr(act, torch.int32),
                                            _ptr(logp_old, torch.float32), _ptr(ent_old, torch.float32),
                                            _ptr(adv, torch.float32), _ptr(ret, torch.float32), C.c_int(B),
                                            C.c_int(A), C.byref(c), _ptr(dlogits_out), _ptr(dvalue_out),
                                            _ptr(metrics_sum, torch.float64, True),
                                            _ptr(workspace, None, True), _stream()),
          "gymrl_ppo_full_loss_fwd_bwd")
    return dlogits_out, dvalue_out


def pack_rollout(obs, act, logp, adv, ret, packed=None):
    """P6: one 64-B record per transition (ppo_lunarlander.py:238-250).  obs [M, D]."""
    M, D = obs.shape
    packed = torch.empty(M, 16, dtype=torch.float32, device=obs.device) if packed is None else packed
    check(lib().gymrl_pack_rollout(_ptr(obs, torch.float32), _ptr(act, torch.int32), _ptr(logp, torch.float32),
                                   _ptr(adv, torch.float32), _ptr(ret, torch.float32), C.c_int64(M), C.c_int(D),
                                   _ptr(packed, torch.float32), _stream()), "gymrl_pack_rollout")
    return packed


def gather_minibatch(packed, idx, obs_dim, out=None):
    """P7: rows idx[0..B) of the packed rollout -> contiguous (obs, act, logp_old, adv, ret)."""
    B, dev = idx.numel(), packed.device
    if out is None:
        out = (torch.empty(B, obs_dim, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
               torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev))
    obs, act, logp, adv, ret = out
    check(lib().gymrl_gather_minibatch(_ptr(packed, torch.float32), _ptr(idx, torch.int32), C.c_int(B),
                                       C.c_int(obs_dim), _ptr(obs, torch.float32), _ptr(act, torch.int32),
                                       _ptr(logp, torch.float32), _ptr(adv, torch.float32),
                                       _ptr(ret, torch.float32), _stream()), "gymrl_gather_minibatch")
    return out


# -------------------------------------------------------------- optimiser ---
def sqnorm(g, out, workspace, grad_scale=1.0):
    check(lib().gymrl_sqnorm(_ptr(g, torch.float32), C.c_int64(g.numel()), C.c_float(grad_scale),
                             _ptr(out, torch.float64), _ptr(workspace), _stream()), "gymrl_sqnorm")
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, max_grad_norm=0.0, sqnorm_buf=None,
              clamp_abs=0.0, zero_grad=True, lr_dev=None):
    """O1: Adam on one flat buffer (ppo_lunarlander.py:302-307); call sqnorm() first when clipping."""
    check(lib().gymrl_adam_step(_ptr(p, torch.float32), _ptr(g, torch.float32), _ptr(m, torch.float32),
                                _ptr(v, torch.float32), C.c_int64(p.numel()), C.c_double(lr),
                                _ptr(lr_dev, torch.float32, True), C.c_double(beta1), C.c_double(beta2),
                                C.c_double(eps), C.c_int64(step), C.c_float(grad_scale),
                                C.c_float(max_grad_norm), _ptr(sqnorm_buf, torch.float64, True),
                                C.c_float(clamp_abs), C.c_int(int(zero_grad)), _stream()), "gymrl_adam_step")


def soft_update(target, source, tau):
    check(lib().gymrl_soft_update(_ptr(target, torch.float32), _ptr(source, torch.float32),
                                  C.c_int64(target.numel()), C.c_double(tau), _stream()), "gymrl_soft_update")


# -------------------------------------------------------------------- env ---
CARTPOLE, PENDULUM, LUNARLANDER = 0, 1, 2
ENV_KINDS = {"CartPole-v1": CARTPOLE, "Pendulum-v1": PENDULUM, "LunarLander-v3": LUNARLANDER}


def env_dims(kind):
    L = lib()
    return (L.gymrl_env_obs_dim(kind), L.gymrl_env_act_dim(kind), bool(L.gymrl_env_is_discrete(kind)),
            L.gymrl_env_max_steps(kind))


def env_state(kind, n, device):
    nbytes = int(lib().gymrl_env_state_bytes(C.c_int(kind), C.c_int(n)))
    if nbytes == 0:
        raise RuntimeError(f"env kind {kind} unavailable")
    return torch.zeros(nbytes, dtype=torch.uint8, device=device)


def env_reset(kind, state, n, seed, env_id0, obs_out):
    check(lib().gymrl_env_reset(C.c_int(kind), _ptr(state), C.c_int(n), C.c_uint64(seed), C.c_int64(env_id0),
                                _ptr(obs_out, torch.float32), _stream()), "gymrl_env_reset")


def env_step(kind, state, n, seed, env_id0, action, obs_out, rew_out, terminated_out, truncated_out,
             term_obs_out=None, done_out=None, ep_ret_out=None, ep_len_out=None, ep_stats=None):
    check(lib().gymrl_env_step(C.c_int(kind), _ptr(state), C.c_int(n), C.c_uint64(seed), C.c_int64(env_id0),
                               _ptr(action), _ptr(obs_out, torch.float32), _ptr(term_obs_out, torch.float32, True),
                               _ptr(rew_out, torch.float32), _ptr(terminated_out, torch.uint8),
                               _ptr(truncated_out, torch.uint8), _ptr(done_out, torch.uint8, True),
                               _ptr(ep_ret_out, torch.float32, True), _ptr(ep_len_out, torch.int32, True),
                               _ptr(ep_stats, torch.float64, True), _stream()), "gymrl_env_step")
