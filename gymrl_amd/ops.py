"""Thin torch-tensor front-end over the C-ABI (include/gymrl.h).

PyTorch is plumbing here: it owns device memory and the stream; every op below
passes raw device pointers + the current HIP stream to libgymrl_hip.so.  No op
has a CPU path — tensors must live on an MI355X.
"""
import collections
import ctypes as C

import torch

from ._lib import (ACT_NONE, ACT_RELU, ACT_TANH, MLP_MAX_INPUT, MLP_MAX_STAGES, MLP_MAX_WIDTH, GaeOnline,
                   MlpDesc, PPOCfg, PPOFullCfg, RainbowActArgs, RainbowUpdateArgs, RolloutLunarArgs, SacActArgs, SacUpdateArgs,
                   check, lib)

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


def gae_decoupled_workspace(T, N, device):
    nbytes = lib().gymrl_gae_decoupled_workspace_bytes(C.c_int(T), C.c_int(N))
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def gae_decoupled(rew, val, done, next_val, gamma, lam_actor, lam_critic, variant=0, workspace=None, adv_out=None,
                  ret_out=None):
    """G3 (ppo_full_lunarlander.py:507-535).  Returns (adv_actor, returns).  variant 1 / 2: time-blocked scan
    (2: chunk maps composed during the rollout), needs gae_decoupled_workspace(T, N)."""
    T, N = rew.shape
    adv = torch.empty_like(rew) if adv_out is None else adv_out
    ret = torch.empty_like(rew) if ret_out is None else ret_out
    if variant and workspace is None:
        workspace = gae_decoupled_workspace(T, N, rew.device)
    check(lib().gymrl_gae_decoupled(_ptr(rew, torch.float32), _ptr(val, torch.float32), _ptr(done, torch.uint8),
                                    _ptr(next_val, torch.float32), C.c_int(T), C.c_int(N), C.c_double(gamma),
                                    C.c_double(lam_actor), C.c_double(lam_critic), _ptr(adv), _ptr(ret),
                                    C.c_int(variant), _ptr(workspace, None, True), _stream()), "gymrl_gae_decoupled")
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
def gae_online(rew_prev, done_prev, val_prev, running, workspace, t_prev, T, gamma, lam, lam2=0.0, running2=None):
    """Descriptor for the producer-side GAE fusion (gymrl_gae_online): rows t-1 of the slab.  lam2 / running2: the
    decoupled-lambda mode of G3 (lam = lam_actor, lam2 = lam_critic)."""
    return GaeOnline(rew_prev.data_ptr(), done_prev.data_ptr(), val_prev.data_ptr(), running.data_ptr(),
                     workspace.data_ptr(), int(t_prev), int(T), float(gamma), float(lam), float(lam2),
                     running2.data_ptr() if running2 is not None else None)


def gae_online_flush(online, val_cur):
    check(lib().gymrl_gae_online_flush(C.byref(online), _ptr(val_cur, torch.float32), C.c_int(val_cur.numel()),
                                       _stream()), "gymrl_gae_online_flush")


def categorical_sample(logits, value=None, noise_exp=None, seed=0, counter=0, env_id0=0, deterministic=False,
                       act_out=None, logp_out=None, ent_out=None, value_out=None, online=None):
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
                                         _ptr(ent_out, None, True), _ptr(value_out, None, True),
                                         C.byref(online) if online is not None else None, _stream()),
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
                          dvalue_out=None, metrics_sum=None, workspace=None, corr_mul=None, entropy_coef_dev=None):
    """L3 (ppo_full_lunarlander.py:575-652).  entropy_coef_dev: f32[1] on the device that overrides cfg's entropy coefficient
    (a captured graph replayed across updates)."""
    B, A = logits.shape
    dlogits_out = torch.empty_like(logits) if dlogits_out is None else dlogits_out
    dvalue_out = torch.empty(B, dtype=torch.float32, device=logits.device) if dvalue_out is None else dvalue_out
    c = PPOFullCfg(*cfg)
    c.entropy_coef_dev = None if entropy_coef_dev is None else _ptr(entropy_coef_dev, torch.float32).value
    if metrics_sum is not None and workspace is None:
        workspace = _reduce_ws(logits.device)
    check(lib().gymrl_ppo_full_loss_fwd_bwd(_ptr(logits, torch.float32), _ptr(value, torch.float32),
                                            _ptr(idx, torch.int32, True), _ptr(act, torch.int32),
                                            _ptr(logp_old, torch.float32), _ptr(ent_old, torch.float32),
                                            _ptr(adv, torch.float32), _ptr(ret, torch.float32), C.c_int(B),
                                            C.c_int(A), C.byref(c), _ptr(corr_mul, torch.float32, True), _ptr(dlogits_out),
                                            _ptr(dvalue_out), _ptr(metrics_sum, torch.float64, True),
                                            _ptr(workspace, None, True), _stream()),
          "gymrl_ppo_full_loss_fwd_bwd")
    return dlogits_out, dvalue_out


def ppo_rnn_loss_fwd_bwd(logits, value, act, logp_old, ent_old, val_old, adv, ret, cfg, idx=None, metrics_sum=None,
                         corr_mul=None):
    """L4 (ppo_lstm_lunarlander.py:716-776): masked means + clipped value loss.  metrics_sum f64[10]."""
    B, A = logits.shape
    dlogits, dvalue = torch.empty_like(logits), torch.empty(B, dtype=torch.float32, device=logits.device)
    c = PPOFullCfg(*cfg)
    check(lib().gymrl_ppo_rnn_loss_fwd_bwd(_ptr(logits, torch.float32), _ptr(value, torch.float32),
                                           _ptr(idx, torch.int32, True), _ptr(act, torch.int32),
                                           _ptr(logp_old, torch.float32), _ptr(ent_old, torch.float32),
                                           _ptr(val_old, torch.float32), _ptr(adv, torch.float32),
                                           _ptr(ret, torch.float32), C.c_int(B), C.c_int(A), C.byref(c),
                                           _ptr(corr_mul, torch.float32, True), _ptr(dlogits),
                                           _ptr(dvalue), _ptr(metrics_sum, torch.float64, True),
                                           _ptr(_reduce_ws(logits.device)), _stream()), "gymrl_ppo_rnn_loss_fwd_bwd")
    return dlogits, dvalue


def gru_cell_fwd(gi, gh, h, out=None):
    B, H = h.shape
    out = torch.empty_like(h) if out is None else out
    check(lib().gymrl_gru_cell_fwd(_ptr(gi, torch.float32), _ptr(gh, torch.float32), _ptr(h, torch.float32), C.c_int(B),
                                   C.c_int(H), _ptr(out, torch.float32), _stream()), "gymrl_gru_cell_fwd")
    return out


def gru_cell_bwd(gi, gh, h, dh_out):
    B, H = h.shape
    dgi, dgh, dh = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
    check(lib().gymrl_gru_cell_bwd(_ptr(gi, torch.float32), _ptr(gh, torch.float32), _ptr(h, torch.float32),
                                   _ptr(dh_out, torch.float32), C.c_int(B), C.c_int(H), _ptr(dgi), _ptr(dgh), _ptr(dh),
                                   _stream()), "gymrl_gru_cell_bwd")
    return dgi, dgh, dh


def rnd_reward(predict, target, rew_inout=None, rnd_out=None):
    B, E = predict.shape
    check(lib().gymrl_rnd_reward(_ptr(predict, torch.float32), _ptr(target, torch.float32), C.c_int(B), C.c_int(E),
                                 _ptr(rew_inout, torch.float32, True), _ptr(rnd_out, torch.float32, True), _stream()),
          "gymrl_rnd_reward")


def loss_blocks(B):
    return int(lib().gymrl_loss_blocks(C.c_int(B)))


def reduce_rows(partials, rows, blocks_per_row, K):
    """[rows, blocks_per_row, K] f64 block partials -> [rows, K] sums (one launch)."""
    out = torch.empty(rows, K, dtype=torch.float64, device=partials.device)
    check(lib().gymrl_reduce_rows(_ptr(partials, torch.float64), C.c_int(rows), C.c_int(blocks_per_row), C.c_int(K),
                                  _ptr(out), _stream()), "gymrl_reduce_rows")
    return out


def permutation(seed, counter, M, device, out=None):
    """P6 epoch shuffle: i32[M] keyed bijection of [0, M) (gymrl_permutation)."""
    out = torch.empty(M, dtype=torch.int32, device=device) if out is None else out
    check(lib().gymrl_permutation(C.c_uint64(seed), C.c_uint64(counter), C.c_int64(M), _ptr(out, torch.int32), _stream()),
          "gymrl_permutation")
    return out


def pack_rollout(obs, act, logp, adv, ret, packed=None):
    """P6: one 64-B record per transition (ppo_lunarlander.py:238-250).  obs [M, D]."""
    M, D = obs.shape
    packed = torch.empty(M, 16, dtype=torch.float32, device=obs.device) if packed is None else packed
    check(lib().gymrl_pack_rollout(_ptr(obs, torch.float32), _ptr(act, torch.int32), _ptr(logp, torch.float32),
                                   _ptr(adv, torch.float32), _ptr(ret, torch.float32), C.c_int64(M), C.c_int(D),
                                   _ptr(packed, torch.float32), _stream()), "gymrl_pack_rollout")
    return packed


def gather_rows(src, idx, out=None):
    """gymrl_gather_rows: src[idx] for a 2-D float32 `src` whose rows are a multiple of 16 bytes (idx int32)."""
    B, D = idx.numel(), src.shape[1]
    out = torch.empty(B, D, device=src.device) if out is None else out
    check(lib().gymrl_gather_rows(_ptr(src, torch.float32), _ptr(idx, torch.int32), C.c_int(B), C.c_int(D), _ptr(out, torch.float32),
                                  _stream()), "gymrl_gather_rows")
    return out


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
              clamp_abs=0.0, zero_grad=True, lr_dev=None, bias_dev=None, polyak_target=None, tau=0.0):
    """O1: Adam on one flat buffer (ppo_lunarlander.py:302-307); call sqnorm() first when clipping."""
    check(lib().gymrl_adam_step(_ptr(p, torch.float32), _ptr(g, torch.float32), _ptr(m, torch.float32),
                                _ptr(v, torch.float32), C.c_int64(p.numel()), C.c_double(lr),
                                _ptr(lr_dev, torch.float32, True), C.c_double(beta1), C.c_double(beta2),
                                C.c_double(eps), C.c_int64(step), _ptr(bias_dev, torch.float32, True),
                                C.c_float(grad_scale),
                                C.c_float(max_grad_norm), _ptr(sqnorm_buf, torch.float64, True),
                                C.c_float(clamp_abs), C.c_int(int(zero_grad)), _ptr(polyak_target, torch.float32, True),
                                C.c_double(tau), _stream()), "gymrl_adam_step")


def clip_adam_step(p, g, m, v, lr, beta1, beta2, eps, step, max_grad_norm, workspace, grad_scale=1.0, sqnorm_out=None,
                   clamp_abs=0.0, zero_grad=True, lr_dev=None, bias_dev=None, polyak_target=None, tau=0.0):
    """gymrl_clip_adam_step: sqnorm() + adam_step() as two launches instead of three (same bits)."""
    check(lib().gymrl_clip_adam_step(_ptr(p, torch.float32), _ptr(g, torch.float32), _ptr(m, torch.float32), _ptr(v, torch.float32),
                                     C.c_int64(p.numel()), C.c_double(lr), _ptr(lr_dev, torch.float32, True), C.c_double(beta1),
                                     C.c_double(beta2), C.c_double(eps), C.c_int64(step), _ptr(bias_dev, torch.float32, True),
                                     C.c_float(grad_scale), C.c_float(max_grad_norm), _ptr(sqnorm_out, torch.float64, True),
                                     C.c_float(clamp_abs), C.c_int(int(zero_grad)), _ptr(polyak_target, torch.float32, True),
                                     C.c_double(tau), _ptr(workspace), _stream()), "gymrl_clip_adam_step")


def adam_bias(lr, beta1, beta2, step):
    """Host arithmetic of Adam's step-dependent scalars -> 4 float32 (gymrl_adam_bias)."""
    out = (C.c_float * 4)()
    check(lib().gymrl_adam_bias(C.c_double(lr), C.c_double(beta1), C.c_double(beta2), C.c_int64(step), out), "gymrl_adam_bias")
    return bytes(out)


def store_scalars(dst, payload):
    """payload: bytes (<= 3840, multiple of 4) -> device tensor `dst`, one launch."""
    if len(payload) > dst.numel() * dst.element_size():
        raise ValueError("payload larger than the destination block")
    buf = C.create_string_buffer(payload, len(payload))
    check(lib().gymrl_store_scalars(_ptr(dst), buf, C.c_int(len(payload)), _stream()), "gymrl_store_scalars")


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


def env_refill(kind, state, n, seed, env_id0, stream=None):
    st = _stream() if stream is None else _vp(stream.cuda_stream)
    check(lib().gymrl_env_refill(C.c_int(kind), _ptr(state), C.c_int(n), C.c_uint64(seed), C.c_int64(env_id0), st),
          "gymrl_env_refill")


def env_step(kind, state, n, seed, env_id0, action, obs_out, rew_out, terminated_out, truncated_out,
             term_obs_out=None, done_out=None, ep_ret_out=None, ep_len_out=None, ep_stats=None):
    check(lib().gymrl_env_step(C.c_int(kind), _ptr(state), C.c_int(n), C.c_uint64(seed), C.c_int64(env_id0),
                               _ptr(action), _ptr(obs_out, torch.float32), _ptr(term_obs_out, torch.float32, True),
                               _ptr(rew_out, torch.float32), _ptr(terminated_out, torch.uint8),
                               _ptr(truncated_out, torch.uint8), _ptr(done_out, torch.uint8, True),
                               _ptr(ep_ret_out, torch.float32, True), _ptr(ep_len_out, torch.int32, True),
                               _ptr(ep_stats, torch.float64, True), _stream()), "gymrl_env_step")


# ============================================================== off-policy ===
def env_abandon(kind, state, n, seed, env_id0, cap, obs_inout, flag_inout=None, ep_ret_out=None, ep_len_out=None,
                ep_stats=None):
    """Start the next episode of every env whose running episode reached `cap` steps (include/gymrl.h)."""
    check(lib().gymrl_env_abandon(C.c_int(kind), _ptr(state), C.c_int(n), C.c_uint64(seed), C.c_int64(env_id0), C.c_int(cap),
                                  _ptr(obs_inout, torch.float32), _ptr(flag_inout, torch.uint8, True),
                                  _ptr(ep_ret_out, torch.float32, True), _ptr(ep_len_out, torch.int32, True),
                                  _ptr(ep_stats, torch.float64, True), _stream()), "gymrl_env_abandon")


def _dev(t):
    """Pointer to a block of per-step scalars on the device (see "Per-step scalars on the device", include/gymrl.h)."""
    return _vp(None) if t is None else _ptr(t)


def replay_append(ring, cursor, src_state, src_action, src_reward, src_next_state, src_flag, cursor_dev=None):
    """D2/A3: write n rows at (cursor + i) % cap.  ring = (state, action_words, reward, next_state, flag)."""
    state, action, reward, next_state, flag = ring
    cap, D = state.shape
    AW = action.shape[1]
    n = src_reward.numel()
    check(lib().gymrl_replay_append(_ptr(state, torch.float32), _ptr(action), _ptr(reward, torch.float32),
                                    _ptr(next_state, torch.float32), _ptr(flag, torch.uint8), C.c_int64(cap),
                                    C.c_int64(cursor), C.c_int(D), C.c_int(AW), C.c_int(n),
                                    _ptr(src_state, torch.float32), _ptr(src_action), _ptr(src_reward, torch.float32),
                                    _ptr(src_next_state, torch.float32), _ptr(src_flag, torch.uint8), _dev(cursor_dev),
                                    _stream()), "gymrl_replay_append")


def replay_gather(ring, idx, action_dtype=torch.int32):
    """rows idx -> (state[B,D], action[B,AW], reward[B], next_state[B,D], flag f32[B])."""
    state, action, reward, next_state, flag = ring
    cap, D = state.shape
    AW = action.shape[1]
    B, dev = idx.numel(), state.device
    out = (torch.empty(B, D, device=dev), torch.empty(B, AW, dtype=action_dtype, device=dev),
           torch.empty(B, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev))
    check(lib().gymrl_replay_gather(_ptr(state, torch.float32), _ptr(action), _ptr(reward, torch.float32),
                                    _ptr(next_state, torch.float32), _ptr(flag, torch.uint8), _ptr(idx, torch.int32),
                                    C.c_int(B), C.c_int(D), C.c_int(AW), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                    _ptr(out[3]), _ptr(out[4]), _stream()), "gymrl_replay_gather")
    return out


def uniform_indices(seed, counter, size, B, device, out=None, dev=None):
    idx = torch.empty(B, dtype=torch.int32, device=device) if out is None else out
    check(lib().gymrl_uniform_indices(C.c_uint64(seed), C.c_uint64(counter), C.c_int64(size), C.c_int(B),
                                      _ptr(idx), _dev(dev), _stream()), "gymrl_uniform_indices")
    return idx


def nstep_push(win, n_steps, pushes, gamma, obs, action, reward, next_obs, terminal, done, ring, cursor, dev=None,
               ep_len=None, max_episode_steps=0):
    """S2.  win = (w_state[n,N,D], w_action i32[n,N], w_reward[n,N], w_next[n,N,D], w_terminal u8, w_done u8).
    Returns True when N rows were emitted into the ring at (cursor + env) % cap."""
    w_state, w_action, w_reward, w_next, w_term, w_done = win
    state, act_w, rew, nxt, flag = ring
    N, D = obs.shape
    rc = lib().gymrl_nstep_push(_ptr(w_state, torch.float32), _ptr(w_action, torch.int32), _ptr(w_reward, torch.float32),
                                _ptr(w_next, torch.float32), _ptr(w_term, torch.uint8), _ptr(w_done, torch.uint8),
                                C.c_int(n_steps), C.c_int64(pushes), C.c_int(N), C.c_int(D), C.c_double(gamma),
                                _ptr(obs, torch.float32), _ptr(action, torch.int32), _ptr(reward, torch.float32),
                                _ptr(next_obs, torch.float32), _ptr(terminal, torch.uint8, True), _ptr(done, torch.uint8),
                                _ptr(state, torch.float32), _ptr(act_w), _ptr(rew, torch.float32),
                                _ptr(nxt, torch.float32), _ptr(flag, torch.uint8), C.c_int64(state.shape[0]),
                                C.c_int64(cursor), _dev(dev), _ptr(ep_len, torch.int32, True), C.c_int(max_episode_steps),
                                _stream())
    if rc < 0:
        check(rc, "gymrl_nstep_push")
    return rc == 1


def per_workspace(B, device):
    return torch.empty(int(lib().gymrl_per_workspace_bytes(C.c_int(B))), dtype=torch.uint8, device=device)


def per_update(tree, cap, B, workspace, idx=None, idx_start=0, idx_is_tree=False, prio=None, prio_scalar_dev=None,
               prio_scalar=0.0, idx_start_dev=None):
    check(lib().gymrl_per_update(_ptr(tree, torch.float64), C.c_int64(cap), _ptr(idx, torch.int32, True),
                                 C.c_int64(idx_start), C.c_int(int(idx_is_tree)), _ptr(prio, torch.float64, True),
                                 _ptr(prio_scalar_dev, torch.float64, True), C.c_double(prio_scalar), C.c_int(B),
                                 _dev(idx_start_dev), _ptr(workspace), _stream()), "gymrl_per_update")


def per_max_leaf(tree, cap, out, workspace):
    check(lib().gymrl_per_max_leaf(_ptr(tree, torch.float64), C.c_int64(cap), _ptr(out, torch.float64),
                                   _ptr(workspace), _stream()), "gymrl_per_max_leaf")
    return out


def per_priorities(td, alpha, eps, clip=0.0, out=None):
    out = torch.empty(td.numel(), dtype=torch.float64, device=td.device) if out is None else out
    check(lib().gymrl_per_priorities(_ptr(td, torch.float32), C.c_int(td.numel()), C.c_double(alpha), C.c_double(eps),
                                     C.c_double(clip), _ptr(out, torch.float64), _stream()), "gymrl_per_priorities")
    return out


PER_TD_MAX_BATCH = 512


def per_update_td(tree, cap, idx, td, alpha, eps, workspace, clip=0.0, max_out=None, ticket=None):
    """gymrl_per_update_td: update_priorities of a sampled batch straight from the TD errors (B <= 512, cap < 2^30); with
    max_out f64[1] + ticket i32[1] (zero) the maximum over the leaves after the update comes out of the same two launches."""
    check(lib().gymrl_per_update_td(_ptr(tree, torch.float64), C.c_int64(cap), _ptr(idx, torch.int32), _ptr(td, torch.float32),
                                    C.c_int(idx.numel()), C.c_double(alpha), C.c_double(eps), C.c_double(clip),
                                    _ptr(max_out, torch.float64, True), _ptr(ticket, torch.int32, True), _ptr(workspace),
                                    _stream()), "gymrl_per_update_td")


def per_sample(tree, cap, B, size, beta, workspace, u=None, seed=0, counter=0, variant_b=False, out=None, dev=None):
    if out is None:
        idx = torch.empty(B, dtype=torch.int32, device=tree.device)
        prio = torch.empty(B, dtype=torch.float64, device=tree.device)
        w = torch.empty(B, dtype=torch.float32, device=tree.device)
    else:
        idx, prio, w = out
    check(lib().gymrl_per_sample(_ptr(tree, torch.float64), C.c_int64(cap), _ptr(u, torch.float64, True),
                                 C.c_uint64(seed), C.c_uint64(counter), C.c_int(B), C.c_int64(size), C.c_double(beta),
                                 C.c_int(int(variant_b)), _ptr(idx), _ptr(prio), _ptr(w), _dev(dev), _ptr(workspace),
                                 _stream()), "gymrl_per_sample")
    return idx, prio, w


def noisy_noise(nin, nout, w_eps, b_eps, eps_in=None, eps_out=None, seed=0, counter=0, counter_dev=None):
    check(lib().gymrl_noisy_noise(_ptr(eps_in, torch.float32, True), _ptr(eps_out, torch.float32, True),
                                  C.c_uint64(seed), C.c_uint64(counter), C.c_int(nin), C.c_int(nout),
                                  _ptr(w_eps, torch.float32), _ptr(b_eps, torch.float32), _dev(counter_dev), _stream()),
          "gymrl_noisy_noise")


def epsilon_greedy(q, epsilon, u=None, seed=0, counter=0, env_id0=0, act_out=None):
    n, A = q.shape
    act_out = torch.empty(n, dtype=torch.int32, device=q.device) if act_out is None else act_out
    check(lib().gymrl_epsilon_greedy(_ptr(q, torch.float32), _ptr(u, torch.float32, True), C.c_uint64(seed),
                                     C.c_uint64(counter), C.c_int64(env_id0), C.c_int(n), C.c_int(A),
                                     C.c_float(epsilon), _ptr(act_out, torch.int32), _stream()), "gymrl_epsilon_greedy")
    return act_out


def dqn_td_loss(q, q_next_target, act, rew, flag, gamma_n, q_next_online=None, w=None, loss_sum=None):
    B, A = q.shape
    td = torch.empty(B, device=q.device)
    dq = torch.empty_like(q)
    ws = _reduce_ws(q.device) if loss_sum is not None else None
    check(lib().gymrl_dqn_td_loss(_ptr(q, torch.float32), _ptr(q_next_online, torch.float32, True),
                                  _ptr(q_next_target, torch.float32), _ptr(act, torch.int32), _ptr(rew, torch.float32),
                                  _ptr(flag, torch.float32), _ptr(w, torch.float32, True), C.c_int(B), C.c_int(A),
                                  C.c_double(gamma_n), _ptr(td), _ptr(dq), _ptr(loss_sum, torch.float64, True),
                                  _ptr(ws, None, True), _stream()), "gymrl_dqn_td_loss")
    return td, dq


def noisy_action(mu, std, bound, eps=None, mode=0, noise_clip=0.0, seed=0, counter=0, out=None):
    """gymrl_noisy_action: Gaussian exploration noise (mode 0, numpy-float64 semantics) or TD3 target-policy
    smoothing (mode 1, torch-float32 semantics) on an action tensor; eps = explicit f64 N(0,1) draws."""
    out = torch.empty_like(mu) if out is None else out
    check(lib().gymrl_noisy_action(_ptr(mu, torch.float32), _ptr(eps, torch.float64, True), C.c_uint64(seed),
                                   C.c_uint64(counter), C.c_int64(mu.numel()), C.c_int(mode), C.c_double(std),
                                   C.c_double(noise_clip), C.c_double(bound), _ptr(out, torch.float32), _stream()),
          "gymrl_noisy_action")
    return out


def mse_loss(q, y, sum_out):
    """One critic's F.mse_loss forward+backward: returns dq; sum_out (f64[1]) += sum (q - y)^2."""
    dq = torch.empty_like(q)
    check(lib().gymrl_mse_loss(_ptr(q, torch.float32), _ptr(y, torch.float32), C.c_int(q.numel()), _ptr(dq),
                               _ptr(sum_out, torch.float64), _ptr(_reduce_ws(q.device)), _stream()), "gymrl_mse_loss")
    return dq


def neg_mean_loss(q, sum_out):
    """Actor loss -mean(q): returns dq = -1/B; sum_out (f64[1]) += sum q."""
    dq = torch.empty_like(q)
    check(lib().gymrl_neg_mean_loss(_ptr(q, torch.float32), C.c_int(q.numel()), _ptr(dq), _ptr(sum_out, torch.float64),
                                    _ptr(_reduce_ws(q.device)), _stream()), "gymrl_neg_mean_loss")
    return dq


def dsac_target(probs_n, q1n, q2n, rew, done, log_alpha, gamma):
    y = torch.empty_like(rew)
    B, A = probs_n.shape
    check(lib().gymrl_dsac_target(_ptr(probs_n, torch.float32), _ptr(q1n, torch.float32), _ptr(q2n, torch.float32),
                                  _ptr(rew, torch.float32), _ptr(done, torch.float32), _ptr(log_alpha, torch.float32),
                                  C.c_int(B), C.c_int(A), C.c_double(gamma), _ptr(y), _stream()), "gymrl_dsac_target")
    return y


def dsac_critic_loss(q1, q2, act, y, sums):
    B, A = q1.shape
    d1, d2 = torch.empty_like(q1), torch.empty_like(q2)
    check(lib().gymrl_dsac_critic_loss(_ptr(q1, torch.float32), _ptr(q2, torch.float32), _ptr(act, torch.int32),
                                       _ptr(y, torch.float32), C.c_int(B), C.c_int(A), _ptr(d1), _ptr(d2),
                                       _ptr(sums, torch.float64), _ptr(_reduce_ws(q1.device)), _stream()),
          "gymrl_dsac_critic_loss")
    return d1, d2


def dsac_actor_loss(probs, q1, q2, log_alpha, sums):
    B, A = probs.shape
    dp = torch.empty_like(probs)
    check(lib().gymrl_dsac_actor_loss(_ptr(probs, torch.float32), _ptr(q1, torch.float32), _ptr(q2, torch.float32),
                                      _ptr(log_alpha, torch.float32), C.c_int(B), C.c_int(A), _ptr(dp),
                                      _ptr(sums, torch.float64), _ptr(_reduce_ws(probs.device)), _stream()),
          "gymrl_dsac_actor_loss")
    return dp


def dsac_alpha_step(log_alpha, m, v, sums, B, target_entropy, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, loss_out=None,
                    bias_dev=None):
    check(lib().gymrl_dsac_alpha_step(_ptr(log_alpha, torch.float32), _ptr(m, torch.float32), _ptr(v, torch.float32),
                                      _ptr(sums, torch.float64), C.c_int(B), C.c_double(target_entropy), C.c_double(lr),
                                      C.c_double(beta1), C.c_double(beta2), C.c_double(eps), C.c_int64(step),
                                      _ptr(bias_dev, torch.float64, True), _ptr(loss_out, torch.float64, True), _stream()),
          "gymrl_dsac_alpha_step")


def sac_sample_fwd(mean, log_std, eps, bound):
    B, A = mean.shape
    action, logp = torch.empty_like(mean), torch.empty(B, device=mean.device)
    check(lib().gymrl_sac_sample_fwd(_ptr(mean, torch.float32), _ptr(log_std, torch.float32), _ptr(eps, torch.float32),
                                     C.c_int(B), C.c_int(A), C.c_float(bound), _ptr(action), _ptr(logp), _stream()),
          "gymrl_sac_sample_fwd")
    return action, logp


def sac_sample_bwd(mean, log_std, eps, d_action, d_logp, bound):
    B, A = mean.shape
    dm, ds = torch.empty_like(mean), torch.empty_like(mean)
    check(lib().gymrl_sac_sample_bwd(_ptr(mean, torch.float32), _ptr(log_std, torch.float32), _ptr(eps, torch.float32),
                                     _ptr(d_action, torch.float32, True), _ptr(d_logp, torch.float32, True), C.c_int(B),
                                     C.c_int(A), C.c_float(bound), _ptr(dm), _ptr(ds), _stream()), "gymrl_sac_sample_bwd")
    return dm, ds


def sac_target(rew, done, q1n, q2n, logp_n, log_alpha, gamma):
    y = torch.empty_like(rew)
    check(lib().gymrl_sac_target(_ptr(rew, torch.float32), _ptr(done, torch.float32), _ptr(q1n, torch.float32),
                                 _ptr(q2n, torch.float32), _ptr(logp_n, torch.float32), _ptr(log_alpha, torch.float64),
                                 C.c_int(rew.numel()), C.c_double(gamma), _ptr(y), _stream()), "gymrl_sac_target")
    return y


def sac_critic_loss(q1, q2, y, sums):
    d1, d2 = torch.empty_like(q1), torch.empty_like(q2)
    check(lib().gymrl_sac_critic_loss(_ptr(q1, torch.float32), _ptr(q2, torch.float32), _ptr(y, torch.float32),
                                      C.c_int(q1.numel()), _ptr(d1), _ptr(d2), _ptr(sums, torch.float64),
                                      _ptr(_reduce_ws(q1.device)), _stream()), "gymrl_sac_critic_loss")
    return d1, d2


def sac_actor_loss(logp, q1, q2, log_alpha, target_entropy, sums):
    dl, d1, d2 = torch.empty_like(logp), torch.empty_like(q1), torch.empty_like(q2)
    check(lib().gymrl_sac_actor_loss(_ptr(logp, torch.float32), _ptr(q1, torch.float32), _ptr(q2, torch.float32),
                                     _ptr(log_alpha, torch.float64), C.c_int(logp.numel()), C.c_double(target_entropy),
                                     _ptr(dl), _ptr(d1), _ptr(d2), _ptr(sums, torch.float64),
                                     _ptr(_reduce_ws(q1.device)), _stream()), "gymrl_sac_actor_loss")
    return dl, d1, d2


def sac_alpha_step(log_alpha, m, v, sums, B, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, loss_out=None, bias_dev=None):
    check(lib().gymrl_sac_alpha_step(_ptr(log_alpha, torch.float64), _ptr(m, torch.float64), _ptr(v, torch.float64),
                                     _ptr(sums, torch.float64), C.c_int(B), C.c_double(lr), C.c_double(beta1),
                                     C.c_double(beta2), C.c_double(eps), C.c_int64(step),
                                     _ptr(bias_dev, torch.float64, True), _ptr(loss_out, torch.float64, True), _stream()),
          "gymrl_sac_alpha_step")


def running_norm(x, stats, update=True, out=None):
    N, D = x.shape
    out = torch.empty_like(x) if out is None else out
    check(lib().gymrl_running_norm(_ptr(x, torch.float32), C.c_int(N), C.c_int(D), _ptr(stats, torch.float64),
                                   C.c_int(int(update)), _ptr(out, torch.float32), _stream()), "gymrl_running_norm")
    return out


def reward_scaling(r, done, gamma, R, stats, out=None):
    out = torch.empty_like(r) if out is None else out
    check(lib().gymrl_reward_scaling(_ptr(r, torch.float32), _ptr(done, torch.uint8, True), C.c_int(r.numel()),
                                     C.c_double(gamma), _ptr(R, torch.float64), _ptr(stats, torch.float64),
                                     _ptr(out, torch.float32), _stream()), "gymrl_reward_scaling")
    return out


# ------------------------------------------------------------ MLP forward ---
def mlp_pack(W, packed=None):
    """gymrl_mlp_pack: nn.Linear weight [out, in] -> MFMA B-operand image (f32 1-D).  Reuses `packed`."""
    out_dim, in_dim = W.shape
    n = lib().gymrl_mlp_packed_floats(C.c_int(out_dim), C.c_int(in_dim))
    if packed is None:
        packed = torch.empty(n, dtype=torch.float32, device=W.device)
    elif packed.numel() != n:
        raise ValueError("packed buffer has the wrong size")
    check(lib().gymrl_mlp_pack(_ptr(W, torch.float32), C.c_int(out_dim), C.c_int(in_dim), _ptr(packed, torch.float32),
                               _stream()), "gymrl_mlp_pack")
    return packed


def mlp_desc(stages):
    """Build the stage table of gymrl_mlp_forward.  stages: list of dicts
    {W (packed by mlp_pack), shape=(out_dim, in_dim), b|None, act, src, dst, out|None}; the tensors must stay
    alive (and in place) for as long as the descriptor is used."""
    if not 0 < len(stages) <= MLP_MAX_STAGES:
        raise ValueError(f"1..{MLP_MAX_STAGES} stages")
    d = MlpDesc()
    d.n_stages = len(stages)
    for i, st in enumerate(stages):
        W, out = st["W"], st.get("out")
        e = d.stage[i]
        e.W, e.b = _ptr(W, torch.float32).value, _ptr(st.get("b"), torch.float32, True).value
        e.out_dim, e.in_dim = st["shape"]
        e.act, e.src, e.dst = int(st.get("act", ACT_NONE)), int(st["src"]), int(st["dst"])
        if e.dst < 0:
            if out is None or out.dim() != 2 or out.shape[1] != e.out_dim:
                raise ValueError("dst == -1 needs an `out` tensor [n_rows, out_dim]")
            e.out, e.out_stride = _ptr(out, torch.float32).value, out.stride(0)
    d._keepalive = list(stages)      # the table holds raw pointers: keep the tensors alive with it
    return d


def mlp_forward(x, desc):
    """gymrl_mlp_forward: the whole Linear(+Tanh|ReLU) chain on x [n_rows, in_dim] in one launch; outputs
    go to the `out` tensors named in the descriptor."""
    n, in_dim = x.shape
    check(lib().gymrl_mlp_forward(_ptr(x, torch.float32), C.c_int(n), C.c_int(in_dim), C.byref(desc), _stream()),
          "gymrl_mlp_forward")


# ------------------------------------------------------ MLP update passes ---
def mlp_train_workspace(C_, D, A, device):
    n = lib().gymrl_mlp_train_workspace_bytes(C.c_int(C_), C.c_int(D), C.c_int(A))
    return torch.empty(n, dtype=torch.uint8, device=device)


def linear_tanh_smallk(x, W, b, out):
    """out = tanh(x W^T + b) for a first layer with in_features in {2,3,4,8} (shared.0 + Tanh)."""
    B, D = x.shape
    check(lib().gymrl_linear_tanh_smallk(_ptr(x, torch.float32), _ptr(W, torch.float32), _ptr(b, torch.float32, True),
                                         C.c_int64(B), C.c_int(D), C.c_int(W.shape[0]), _ptr(out, torch.float32),
                                         _stream()), "gymrl_linear_tanh_smallk")
    return out


def linear_smallk(x, W, b, out):
    """out = x W^T + b for a first layer with in_features in {2,3,4,8} and a power-of-two width (no activation)."""
    B, D = x.shape
    check(lib().gymrl_linear_smallk(_ptr(x, torch.float32), _ptr(W, torch.float32), _ptr(b, torch.float32, True),
                                    C.c_int64(B), C.c_int(D), C.c_int(W.shape[0]), _ptr(out, torch.float32), _stream()),
          "gymrl_linear_smallk")
    return out


def tanh_inplace(z, bias=None):
    """z <- tanh(z + bias) in place; bias [C] broadcasts over the rows of z [..., C]."""
    check(lib().gymrl_tanh_inplace(_ptr(z, torch.float32), C.c_int64(z.numel()), _ptr(bias, torch.float32, True),
                                   C.c_int(z.shape[-1]), _stream()), "gymrl_tanh_inplace")
    return z


def tanh_bwd_colsum(dH, H, colsum_out, workspace):
    """dH <- dH * (1 - H^2) in place; colsum_out[c] = sum_r dH[r, c]."""
    B, Cc = dH.shape
    check(lib().gymrl_tanh_bwd_colsum(_ptr(dH, torch.float32), _ptr(H, torch.float32), C.c_int64(B), C.c_int(Cc),
                                      _ptr(colsum_out, torch.float32), _ptr(workspace), _stream()),
          "gymrl_tanh_bwd_colsum")
    return dH


def linear_smallk_bwd(dH, H, x, dW, db, workspace, W=None, b=None):
    """First layer backward without materialising dZ: dW = (dH (1 - H^2))^T x, db = colsum.  With the layer's own
    (W, b) the kernel recomputes H = tanh(x W^T + b) instead of reading it (H may then be None).  dW = db = None: the
    block partials only (they stay in `workspace` for update_finalize)."""
    B, Cc = dH.shape
    check(lib().gymrl_linear_smallk_bwd(_ptr(dH, torch.float32), _ptr(H, torch.float32, True), _ptr(x, torch.float32),
                                        C.c_int64(B), C.c_int(x.shape[1]), C.c_int(Cc), _ptr(dW, torch.float32, True),
                                        _ptr(db, torch.float32, True), _ptr(W, torch.float32, True), _ptr(b, torch.float32, True),
                                        _ptr(workspace), _stream()), "gymrl_linear_smallk_bwd")


def heads_fwd_tanh(Zac, Wa2, ba2, Wc2, bc2, logits, value, bac=None, store_h=True):
    """Zac [B, 2C] -> tanh in place + logits [B, A] + value [B] (include/gymrl.h gymrl_heads_fwd_tanh)."""
    B, C2 = Zac.shape
    check(lib().gymrl_heads_fwd_tanh(_ptr(Zac, torch.float32), C.c_int64(B), C.c_int(C2 // 2), C.c_int(Wa2.shape[0]),
                                     _ptr(bac, torch.float32, True), _ptr(Wa2, torch.float32), _ptr(ba2, torch.float32, True), _ptr(Wc2, torch.float32),
                                     _ptr(bc2, torch.float32, True), _ptr(logits, torch.float32),
                                     _ptr(value, torch.float32), C.c_int(int(store_h)), _stream()), "gymrl_heads_fwd_tanh")


def heads_bwd(Hac, dlogits, dv, Wa2, Wc2, dZac, dbac, dWa2, dba2, dWc2, dbc2, workspace, pre_activation=False, bac=None):
    """Backward of both heads in one pass over Hac = [Ha | Hc] (include/gymrl.h gymrl_heads_bwd)."""
    B, C2 = Hac.shape
    check(lib().gymrl_heads_bwd(_ptr(Hac, torch.float32), _ptr(dlogits, torch.float32), _ptr(dv, torch.float32),
                                C.c_int64(B), C.c_int(C2 // 2), C.c_int(dlogits.shape[1]), _ptr(Wa2, torch.float32),
                                _ptr(Wc2, torch.float32), _ptr(dZac, torch.float32), _ptr(dbac, torch.float32),
                                _ptr(dWa2, torch.float32), _ptr(dba2, torch.float32), _ptr(dWc2, torch.float32),
                                _ptr(dbc2, torch.float32), C.c_int(int(pre_activation)), _ptr(bac, torch.float32, True),
                                _ptr(workspace), _stream()), "gymrl_heads_bwd")


def heads_loss_blocks(B, C_=256):
    return int(lib().gymrl_heads_loss_blocks(C.c_int64(B), C.c_int(C_)))


def heads_loss_fwd_bwd(Zac, bac, Wa2, ba2, Wc2, bc2, act, logp_old, adv, ret, cfg, adv_moments, dbac, dWa2, dba2, dWc2,
                       dbc2, metric_parts, workspace):
    """Heads forward + PPO loss + heads backward in one pass; dZac overwrites Zac (include/gymrl.h).  All five gradient
    outputs None: the block partials only (they stay in `workspace` for update_finalize)."""
    B, C2 = Zac.shape
    c = PPOCfg(*[float(v) for v in cfg])
    check(lib().gymrl_heads_loss_fwd_bwd(
        _ptr(Zac, torch.float32), C.c_int64(B), C.c_int(C2 // 2), C.c_int(Wa2.shape[0]), _ptr(bac, torch.float32, True),
        _ptr(Wa2, torch.float32), _ptr(ba2, torch.float32, True), _ptr(Wc2, torch.float32), _ptr(bc2, torch.float32, True),
        _ptr(act, torch.int32), _ptr(logp_old, torch.float32), _ptr(adv, torch.float32), _ptr(ret, torch.float32),
        _ptr(adv_moments, torch.float64, True), C.byref(c), _ptr(dbac, torch.float32, True), _ptr(dWa2, torch.float32, True),
        _ptr(dba2, torch.float32, True), _ptr(dWc2, torch.float32, True), _ptr(dbc2, torch.float32, True), _ptr(metric_parts, torch.float64),
        _ptr(workspace), _stream()), "gymrl_heads_loss_fwd_bwd")


# ------------------------------------------------------ update-path GEMMs ---
def gemm_workspace(device):
    return torch.empty(int(lib().gymrl_gemm_workspace_bytes()), dtype=torch.uint8, device=device)


def gemm_config(key, value):
    """Ablation variants of the hand-written GEMMs — probe build only (make -C gymrl_amd/csrc prof, GYMRL_HIP_LIB=
    .../libgymrl_hip_prof.so); the product library does not export gymrl_gemm_config."""
    L = lib()
    if not hasattr(L, "gymrl_gemm_config"):
        raise RuntimeError("gymrl_gemm_config exists in the probe build only (make -C gymrl_amd/csrc prof)")
    check(L.gymrl_gemm_config(C.c_int(key), C.c_int(value)), "gymrl_gemm_config")


def linear_shape_ok(K, N):
    """Shapes gymrl_linear_fwd (x [B, K] -> [B, N]) and gymrl_linear_bwd_input (dy [B, N] -> [B, K]) cover: K in
    {64, 128} with N in {K, 2K}; K = 256 with N in {256, 512}."""
    return (K in (64, 128) and N in (K, 2 * K)) or (K == 256 and N in (256, 512))


def linear_fwd(x, W, b, out, act=True):
    """out [B, N] = tanh(x [B, K] W[N, K]^T + b) (act=False: no tanh; b None: no bias) — exact-f32 MFMA, fused epilogue."""
    B, K = x.shape
    check(lib().gymrl_linear_fwd(_ptr(x, torch.float32), _ptr(W, torch.float32), _ptr(b, torch.float32, True),
                                 C.c_int64(B), C.c_int(K), C.c_int(W.shape[0]), C.c_int(int(act)), _ptr(out, torch.float32),
                                 _stream()), "gymrl_linear_fwd")
    return out


def linear_bwd_input(dy, W, H, dx):
    """dx [B, K] = (dy [B, N] W[N, K]) * (1 - H^2)  (H None: no factor)."""
    B, N = dy.shape
    check(lib().gymrl_linear_bwd_input(_ptr(dy, torch.float32), _ptr(W, torch.float32), _ptr(H, torch.float32, True),
                                       C.c_int64(B), C.c_int(N), C.c_int(W.shape[1]), _ptr(dx, torch.float32), _stream()),
          "gymrl_linear_bwd_input")
    return dx


def linear_bwd_input_add(dy, W, g, dx):
    """dx [B, 128] = g + dy [B, 256] W[256, 128]: a second Linear's input gradient added to the first one's in the epilogue."""
    B, N = dy.shape
    if g.shape != dx.shape or g.data_ptr() == dx.data_ptr():
        raise ValueError("linear_bwd_input_add: g and dx are two tensors of one shape")
    check(lib().gymrl_linear_bwd_input_add(_ptr(dy, torch.float32), _ptr(W, torch.float32), _ptr(g, torch.float32),
                                           C.c_int64(B), C.c_int(N), C.c_int(W.shape[1]), _ptr(dx, torch.float32), _stream()),
          "gymrl_linear_bwd_input_add")
    return dx


def linear_bwd_weight_geometry(B, N):
    s, r = C.c_int(0), C.c_int64(0)
    check(lib().gymrl_linear_bwd_weight_geometry(C.c_int64(B), C.c_int(N), C.byref(s), C.byref(r)),
          "gymrl_linear_bwd_weight_geometry")
    return s.value, r.value


def linear_bwd_weight(dy, x, dW, workspace, db=None, partials_db=False):
    """dW [N, 256] = dy [B, N]^T x [B, 256]; db [N] = column sums of dy (None: skipped).  Overwrites.  dW = None: the slice
    partials only (they stay in `workspace` for update_finalize; partials_db: with the column-sum partials)."""
    B, N = dy.shape
    if dW is None and partials_db:
        db_arg = _ptr(workspace)                       # a non-NULL flag: nothing is written through it in this mode
    else:
        db_arg = _ptr(db, torch.float32, True)
    check(lib().gymrl_linear_bwd_weight(_ptr(dy, torch.float32), _ptr(x, torch.float32), C.c_int64(B), C.c_int(N),
                                        C.c_int(x.shape[1]), _ptr(dW, torch.float32, True), db_arg,
                                        _ptr(workspace), _stream()), "gymrl_linear_bwd_weight")
    return dW


def update_finalize(B, C_, A, D, ws_dw_ac, dWac, ws_dw_2, dW2, db2, ws_heads, dbac, dWa2, dba2, dWc2, dbc2, ws_smallk, dW1, db1):
    """gymrl_update_finalize: the second halves of a minibatch's five batch reductions (both 256-deep weight gradients, the
    heads' and the first layer's gradients) as ONE launch, from the partials their producers left in their workspaces."""
    f = lambda t: _ptr(t, torch.float32)        # noqa: E731
    check(lib().gymrl_update_finalize(C.c_int64(B), C.c_int(C_), C.c_int(A), C.c_int(D), _ptr(ws_dw_ac), f(dWac), _ptr(ws_dw_2),
                                      f(dW2), f(db2), _ptr(ws_heads), f(dbac), f(dWa2), f(dba2), f(dWc2), f(dbc2), _ptr(ws_smallk),
                                      f(dW1), f(db1), _stream()), "gymrl_update_finalize")


# ------------------------------------------------------ persistent rollout ---
def rollout_lunar(env_state, n_envs, seed, env_id0, counter0, obs, act, logp, val, rew, done, ep_ret, next_value, policy_desc,
                  T, t0, nsteps, gamma, lam, noise_exp=None, gae_running=None, gae_workspace=None, ep_stats=None, wg_ticks=None,
                  refill=True, gae_carry=False):
    """gymrl_rollout_lunar: `nsteps` vector steps of collect_rollout (policy forward, draw, env step, slab writes,
    online GAE) in one launch; see include/gymrl.h for the slab layout.  gae_carry: the launch that reaches T also runs the
    blocked scan's carry pass for its envs (then gae(..., variant=3) is the apply launch alone)."""
    a = RolloutLunarArgs()
    a.env_state, a.n_envs, a.seed, a.env_id0, a.counter0 = _ptr(env_state).value, n_envs, seed, env_id0, counter0
    a.obs, a.act, a.logp = _ptr(obs, torch.float32).value, _ptr(act, torch.int32).value, _ptr(logp, torch.float32).value
    a.val, a.rew, a.done = _ptr(val, torch.float32).value, _ptr(rew, torch.float32).value, _ptr(done, torch.uint8).value
    a.ep_ret, a.next_value = _ptr(ep_ret, torch.float32, True).value, _ptr(next_value, torch.float32).value
    a.noise_exp = _ptr(noise_exp, torch.float32, True).value
    a.gae_running, a.gae_workspace = _ptr(gae_running, torch.float64, True).value, _ptr(gae_workspace, None, True).value
    a.gamma, a.lam, a.ep_stats = gamma, lam, _ptr(ep_stats, torch.float64, True).value
    a.wg_ticks = _ptr(wg_ticks, torch.int64, True).value
    a.T, a.t0, a.nsteps, a.refill = T, t0, nsteps, int(bool(refill))
    a.gae_carry = int(bool(gae_carry) and gae_running is not None)
    check(lib().gymrl_rollout_lunar(C.byref(a), C.byref(policy_desc), _stream()), "gymrl_rollout_lunar")


def rollout_cartpole(env_state, n_envs, seed, env_id0, counter0, obs, act, logp, val, rew, done, ep_ret, next_value, policy_desc,
                     T, t0, nsteps, gamma, lam, noise_exp=None, gae_running=None, gae_workspace=None, ep_stats=None):
    """gymrl_rollout_cartpole: the persistent rollout for PPO on CartPole-v1 (obs [T+1, N, 4], two actions)."""
    a = RolloutLunarArgs()
    a.env_state, a.n_envs, a.seed, a.env_id0, a.counter0 = _ptr(env_state).value, n_envs, seed, env_id0, counter0
    a.obs, a.act, a.logp = _ptr(obs, torch.float32).value, _ptr(act, torch.int32).value, _ptr(logp, torch.float32).value
    a.val, a.rew, a.done = _ptr(val, torch.float32).value, _ptr(rew, torch.float32).value, _ptr(done, torch.uint8).value
    a.ep_ret, a.next_value = _ptr(ep_ret, torch.float32, True).value, _ptr(next_value, torch.float32).value
    a.noise_exp = _ptr(noise_exp, torch.float32, True).value
    a.gae_running, a.gae_workspace = _ptr(gae_running, torch.float64, True).value, _ptr(gae_workspace, None, True).value
    a.gamma, a.lam, a.ep_stats = gamma, lam, _ptr(ep_stats, torch.float64, True).value
    a.T, a.t0, a.nsteps = T, t0, nsteps
    check(lib().gymrl_rollout_cartpole(C.byref(a), C.byref(policy_desc), _stream()), "gymrl_rollout_cartpole")


def rollout_lunar_mhc(env_state, n_envs, seed, env_id0, counter0, obs, act, logp, val, rew, done, ep_ret, next_value, policy_desc,
                      T, t0, nsteps, gamma, lam, ent=None, lam2=0.0, noise_exp=None, gae_running=None, gae_running2=None,
                      gae_workspace=None, ep_stats=None, refill=True):
    """gymrl_rollout_lunar_mhc: the persistent LunarLander rollout with PPO-full's mHC network (a filled _lib.MhcPolicy) as the
    policy; `ent` [T, N] receives the behaviour entropies, gae_running2 / lam2 switch the decoupled-lambda maps on."""
    a = RolloutLunarArgs()
    a.env_state, a.n_envs, a.seed, a.env_id0, a.counter0 = _ptr(env_state).value, n_envs, seed, env_id0, counter0
    a.obs, a.act, a.logp = _ptr(obs, torch.float32).value, _ptr(act, torch.int32).value, _ptr(logp, torch.float32).value
    a.val, a.rew, a.done = _ptr(val, torch.float32).value, _ptr(rew, torch.float32).value, _ptr(done, torch.uint8).value
    a.ep_ret, a.next_value = _ptr(ep_ret, torch.float32, True).value, _ptr(next_value, torch.float32).value
    a.noise_exp = _ptr(noise_exp, torch.float32, True).value
    a.gae_running, a.gae_workspace = _ptr(gae_running, torch.float64, True).value, _ptr(gae_workspace, None, True).value
    a.gae_running2, a.lam2, a.ent = _ptr(gae_running2, torch.float64, True).value, lam2, _ptr(ent, torch.float32, True).value
    a.gamma, a.lam, a.ep_stats = gamma, lam, _ptr(ep_stats, torch.float64, True).value
    a.T, a.t0, a.nsteps, a.refill = T, t0, nsteps, int(bool(refill))
    check(lib().gymrl_rollout_lunar_mhc(C.byref(a), C.byref(policy_desc), _stream()), "gymrl_rollout_lunar_mhc")


# ------------------------------------------------ low-latency Linear layers --
LIN_ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2, "clamp": 3, "dueling": 4, "silu": 5}
LIN_MAX_ITEMS = 4


def _rows(t, allow_none=False):
    """(pointer, row stride) of a 2-D f32 device tensor whose rows are contiguous (column slices are fine)."""
    if t is None:
        if allow_none:
            return None, 0
        raise ValueError("tensor required")
    if not t.is_cuda:
        raise RuntimeError("gymrl_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
    if t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("expected a 2-D float32 tensor with contiguous rows")
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _as_items(v, n):
    if isinstance(v, (list, tuple)):
        if len(v) != n:
            raise ValueError("every per-item argument needs the same number of items")
        return list(v)
    return [v] * n


def _lin_pack(n, **cols):
    """Build the gymrl_lin_item array; cols: field -> list of (ptr | None).  Returns the ctypes array."""
    from ._lib import LinItem
    arr = (LinItem * n)()
    for f, vals in cols.items():
        for i, v in enumerate(vals):
            setattr(arr[i], f, v)
    return arr


def _same(vals, what):
    if any(v != vals[0] for v in vals):
        raise ValueError(f"items of one launch must share {what}")
    return vals[0]


def lin_workspace(B, N, K, n_items, device):
    n = lib().gymrl_lin_workspace_bytes(C.c_int(B), C.c_int(N), C.c_int(K), C.c_int(n_items))
    return torch.empty(max(n // 4, 1), dtype=torch.float32, device=device) if n else None


def lin_fwd(x, w, b, act=0, x2=None, out=None, lo=0.0, hi=0.0, argmax=None):
    """gymrl_lin_fwd: y = act(cat(x, x2) w^T + b) in one launch.  Every tensor argument may be a list (<= 4 layers of one
    shape in the same launch: twin critics, two heads on one input); returns y or the list of ys."""
    multi = isinstance(w, (list, tuple))
    n = len(w) if multi else 1
    xs, x2s, ws, bs = _as_items(x, n), _as_items(x2, n), _as_items(w, n), _as_items(b, n)
    B = xs[0].shape[0]
    N, K = ws[0].shape
    K1 = xs[0].shape[1]
    acts = _as_items(act, n)
    outs = _as_items(out, n) if out is not None else [torch.empty(B, N - 1 if a == 4 else N, dtype=torch.float32,
                                                                  device=xs[0].device) for a in acts]
    px, ldx = zip(*(_rows(t) for t in xs))
    px2, ldx2 = zip(*(_rows(t, True) for t in x2s))
    py, ldy = zip(*(_rows(t) for t in outs))
    for wt, xt, x2t in zip(ws, xs, x2s):
        if tuple(wt.shape) != (N, K) or xt.shape != (B, K1) or K1 + (0 if x2t is None else x2t.shape[1]) != K:
            raise ValueError("lin_fwd: shape mismatch between the items of one launch")
    items = _lin_pack(n, x=px, x2=px2, w=[_ptr(t, torch.float32).value for t in ws],
                      b=[None if t is None else _ptr(t, torch.float32).value for t in bs], y=py,
                      act=acts, lo=_as_items(lo, n), hi=_as_items(hi, n),
                      argmax=[None if t is None else _ptr(t, torch.int32).value for t in _as_items(argmax, n)])
    check(lib().gymrl_lin_fwd(items, C.c_int(n), C.c_int(B), C.c_int(K), C.c_int(K1), C.c_int(N),
                              C.c_int(_same(ldx, "a row stride")), C.c_int(_same(ldx2, "a row stride")),
                              C.c_int(_same(ldy, "a row stride")), _stream()), "gymrl_lin_fwd")
    return outs if multi else outs[0]


def lin_bwd_input(dy, y, w, act=0, K1=None, dx=None, dx2=None, want=(True, True), lo=0.0, hi=0.0, accumulate=False,
                  sum_items=False):
    """gymrl_lin_bwd_input: (dx | dx2) = (dy * act'(y)) w.  K1 = columns of the first input block (None: all of K);
    want = which of the two blocks to compute.  Lists batch up to 4 layers; sum_items: the layers share their input and
    ONE gradient (the sum) is returned.  Returns (dx, dx2) (None where skipped)."""
    multi = isinstance(w, (list, tuple))
    n = len(w) if multi else 1
    dys, ys, ws = _as_items(dy, n), _as_items(y, n), _as_items(w, n)
    B, N = dys[0].shape
    K = ws[0].shape[1]
    K1 = K if K1 is None else K1
    dev = dys[0].device
    want1, want2 = want[0] and K1 > 0, want[1] and K1 < K
    n_out = 1 if sum_items else n
    dxs = _as_items(dx, n) if dx is not None else [torch.empty(B, K1, dtype=torch.float32, device=dev)
                                                   if want1 and i < n_out else None for i in range(n)]
    dx2s = _as_items(dx2, n) if dx2 is not None else [torch.empty(B, K - K1, dtype=torch.float32, device=dev)
                                                      if want2 and i < n_out else None for i in range(n)]
    if sum_items:                       # the C side reads item 0's destinations only; keep the strides uniform
        dxs, dx2s = [dxs[0]] * n, [dx2s[0]] * n
    pdy, ldy = zip(*(_rows(t) for t in dys))
    acts = _as_items(act, n)
    py, ldyy = zip(*(_rows(t, a == 0) for t, a in zip(ys, acts)))
    if any(a != 0 and l1 != l2 for a, l1, l2 in zip(acts, ldyy, ldy)):
        raise ValueError("lin_bwd_input: dy and y need the same row stride")
    pdx, lddx = zip(*(_rows(t, True) for t in dxs))
    pdx2, lddx2 = zip(*(_rows(t, True) for t in dx2s))
    items = _lin_pack(n, dy=pdy, y=py, w=[_ptr(t, torch.float32).value for t in ws], dx=pdx, dx2=pdx2,
                      act=acts, lo=_as_items(lo, n), hi=_as_items(hi, n))
    check(lib().gymrl_lin_bwd_input(items, C.c_int(n), C.c_int(B), C.c_int(N), C.c_int(K), C.c_int(K1),
                                    C.c_int(_same(ldy, "a row stride")), C.c_int(_same(lddx, "a row stride")),
                                    C.c_int(_same(lddx2, "a row stride")), C.c_int(int(accumulate)),
                                    C.c_int(int(sum_items)), _stream()), "gymrl_lin_bwd_input")
    if sum_items:
        return dxs[0], dx2s[0]
    return (dxs, dx2s) if multi else (dxs[0], dx2s[0])


def lin_bwd_weight(dy, y, x, dw, db=None, act=0, x2=None, lo=0.0, hi=0.0, accumulate=False, workspace=None):
    """gymrl_lin_bwd_weight: dw (+)= (dy * act'(y))^T cat(x, x2), db (+)= its column sums, written straight into dw / db
    (e.g. the views of a flat gradient buffer).  Lists batch up to 4 layers."""
    multi = isinstance(dw, (list, tuple))
    n = len(dw) if multi else 1
    dys, ys, xs, x2s, dws, dbs = (_as_items(v, n) for v in (dy, y, x, x2, dw, db))
    B, N = dys[0].shape
    K1 = xs[0].shape[1]
    K = dws[0].shape[1]
    pdy, ldy = zip(*(_rows(t) for t in dys))
    acts = _as_items(act, n)
    py, ldyy = zip(*(_rows(t, a == 0) for t, a in zip(ys, acts)))
    if any(a != 0 and l1 != l2 for a, l1, l2 in zip(acts, ldyy, ldy)):
        raise ValueError("lin_bwd_weight: dy and y need the same row stride")
    px, ldx = zip(*(_rows(t) for t in xs))
    px2, ldx2 = zip(*(_rows(t, True) for t in x2s))
    if workspace is None and B > 512:
        workspace = lin_workspace(B, N, K, n, dys[0].device)
    items = _lin_pack(n, dy=pdy, y=py, x=px, x2=px2, dw=[_ptr(t, torch.float32).value for t in dws],
                      db=[None if t is None else _ptr(t, torch.float32).value for t in dbs],
                      act=acts, lo=_as_items(lo, n), hi=_as_items(hi, n))
    check(lib().gymrl_lin_bwd_weight(items, C.c_int(n), C.c_int(B), C.c_int(N), C.c_int(K), C.c_int(K1),
                                     C.c_int(_same(ldy, "a row stride")), C.c_int(_same(ldx, "a row stride")),
                                     C.c_int(_same(ldx2, "a row stride")), C.c_int(int(accumulate)),
                                     _ptr(workspace, torch.float32, True), _stream()),
          "gymrl_lin_bwd_weight")
    return dw


def _noisy_layers(layers):
    """layers: list of dicts with w_mu, w_sigma, w_eps, b_mu, b_sigma, b_eps (+ optional w_eps_copy, b_eps_copy, dw_mu,
    dw_sigma, db_mu, db_sigma) tensors -> (ctypes array, K, total rows)."""
    from ._lib import NoisyLayer
    arr = (NoisyLayer * len(layers))()
    K = layers[0]["w_mu"].shape[1]
    rows = 0
    for i, L in enumerate(layers):
        if L["w_mu"].shape[1] != K:
            raise ValueError("noisy layers of one launch share their input width")
        for f, ct in NoisyLayer._fields_:
            if ct is C.c_void_p and f != "counter_dev":
                t = L.get(f)
                setattr(arr[i], f, None if t is None else _ptr(t, torch.float32).value)
        arr[i].seed, arr[i].counter, arr[i].draw = int(L.get("seed", 0)), int(L.get("counter", 0)), int(bool(L.get("draw")))
        arr[i].eval = int(bool(L.get("eval")))
        cd = L.get("counter_dev")
        arr[i].counter_dev = None if cd is None else _ptr(cd).value
        arr[i].n_out = L["w_mu"].shape[0]
        rows += L["w_mu"].shape[0]
    return arr, K, rows


def noisy_combine(layers, training=True, images=None):
    """gymrl_noisy_combine: stacked effective parameters (W [rows, K], b [rows]) of NoisyLinear layers in one launch.
    images: [(W [H, H], img_fwd or None, img_bwd or None)] — gymrl_noisy_combine_images: the MFMA-operand images of square
    weights rebuilt on extra workgroups of the same launch (flat f32[H * H] outputs)."""
    arr, K, rows = _noisy_layers(layers)
    dev = layers[0]["w_mu"].device
    W, b = torch.empty(rows, K, dtype=torch.float32, device=dev), torch.empty(rows, dtype=torch.float32, device=dev)
    if images:
        from ._lib import WeightImage
        ims = (WeightImage * len(images))()
        for i, (w, f, bw) in enumerate(images):
            H = w.shape[0]
            if w.shape != (H, H) or not w.is_contiguous() or any(t is not None and t.numel() != H * H for t in (f, bw)):
                raise ValueError("noisy_combine: an image needs a contiguous square weight and H * H floats per image")
            ims[i].W, ims[i].H = _ptr(w, torch.float32).value, H
            ims[i].img_fwd, ims[i].img_bwd = (None if t is None else _ptr(t, torch.float32).value for t in (f, bw))
        check(lib().gymrl_noisy_combine_images(arr, C.c_int(len(layers)), C.c_int(K), C.c_int(int(training)), _ptr(W), _ptr(b),
                                               ims, C.c_int(len(images)), _stream()), "gymrl_noisy_combine_images")
        return W, b
    check(lib().gymrl_noisy_combine(arr, C.c_int(len(layers)), C.c_int(K), C.c_int(int(training)), _ptr(W), _ptr(b), _stream()),
          "gymrl_noisy_combine")
    return W, b


def noisy_split(layers, dW, db, training=True, accumulate=False):
    """gymrl_noisy_split: the stacked gradient back to every layer's dw_mu / dw_sigma / db_mu / db_sigma tensors."""
    arr, K, rows = _noisy_layers(layers)
    if tuple(dW.shape) != (rows, K):
        raise ValueError("noisy_split: stacked gradient shape mismatch")
    check(lib().gymrl_noisy_split(arr, C.c_int(len(layers)), C.c_int(K), C.c_int(int(training)), _ptr(dW, torch.float32),
                                  _ptr(db, torch.float32), C.c_int(int(accumulate)), _stream()), "gymrl_noisy_split")


def dueling_bwd(dq):
    """gymrl_dueling_bwd: dq [B, A] -> dS [B, A + 1] (advantage columns, then the value column)."""
    B, A = dq.shape
    dS = torch.empty(B, A + 1, dtype=torch.float32, device=dq.device)
    check(lib().gymrl_dueling_bwd(_ptr(dq, torch.float32), C.c_int(B), C.c_int(A), _ptr(dS), _stream()), "gymrl_dueling_bwd")
    return dS



# ------------------------------------------------ mHC backbone (inference) --
def mhc_gates(h, norm_w, w, alpha, beta, sk_it, stats=False):
    """gymrl_mhc_gates: h [B, n, D] -> (pre [B, n], post [B, n], mix [B, n, n], read [B, D]); stats=True appends the per-row
    read-out sums f32[B, n*n + 2n + 1] gymrl_mhc_gates_bwd takes (n = 2, n*D in (256, 512))."""
    B, n, D = h.shape
    dev = h.device
    pre, post = torch.empty(B, n, device=dev), torch.empty(B, n, device=dev)
    mix, read = torch.empty(B, n, n, device=dev), torch.empty(B, D, device=dev)
    st = torch.empty(B, n * n + 2 * n + 1, device=dev) if stats else None
    check(lib().gymrl_mhc_gates(_ptr(h, torch.float32), _ptr(norm_w, torch.float32), _ptr(w, torch.float32),
                                _ptr(alpha, torch.float32), _ptr(beta, torch.float32), C.c_int(B), C.c_int(n), C.c_int(D),
                                C.c_int(sk_it), _ptr(pre), _ptr(post), _ptr(mix), _ptr(read), None if st is None else _ptr(st),
                                _stream()), "gymrl_mhc_gates")
    return (pre, post, mix, read, st) if stats else (pre, post, mix, read)


def mhc_combine(post, mix, out, h, act=0):
    """gymrl_mhc_combine: h'[b, i] = post[b, i] act(out[b]) + sum_j mix[b, i, j] h[b, j]; act = 0 or LIN_ACT["silu"]."""
    B, n, D = h.shape
    h_out = torch.empty_like(h)
    check(lib().gymrl_mhc_combine(_ptr(post, torch.float32), _ptr(mix, torch.float32), _ptr(out, torch.float32),
                                  _ptr(h, torch.float32), C.c_int(B), C.c_int(n), C.c_int(D), C.c_int(act), _ptr(h_out),
                                  _stream()), "gymrl_mhc_combine")
    return h_out


def rmsnorm(x, w, eps, n_sum=1, act=0):
    """gymrl_rmsnorm: x [B, n_sum * D] (or [B, n_sum, D]) -> y [B, D] = s rsqrt(mean(s^2) + eps) w, s = act(the sum of the blocks)."""
    B, D = x.shape[0], w.numel()
    y = torch.empty(B, D, device=x.device)
    check(lib().gymrl_rmsnorm(_ptr(x, torch.float32), _ptr(w, torch.float32), C.c_int(B), C.c_int(D), C.c_int(n_sum),
                              C.c_float(eps), C.c_int(act), _ptr(y), _stream()), "gymrl_rmsnorm")
    return y


_SCRATCH_MAX = 64
_scratch_cache = collections.OrderedDict()


def _scratch(kind, shape_key, nbytes, dev):
    """Partial-sum scratch of the backward kernels that take one.  Cached per (kernel, shape, device, STREAM): two
    streams never share a buffer, so concurrent backward passes cannot race on it.  While the stream is capturing a
    hipGraph a missing buffer is allocated for this call only (it then lives in the graph's private pool like any other
    temporary) and is NOT cached — an eager call after `del graph` must never inherit memory of a dead graph's pool.
    The cache is a bounded LRU (_SCRATCH_MAX entries): code that keeps creating streams cannot grow it without limit, and
    an evicted buffer goes back to torch's caching allocator, which hands a block to another stream only after the work
    queued on the allocating stream (the one in the key) has been ordered before the reuse — so a raw stream handle that
    is recycled after its stream died finds either its own old buffer (same stream-ordering domain) or none."""
    key = (kind, shape_key, dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _scratch_cache.get(key)
    if ws is not None:
        _scratch_cache.move_to_end(key)
        return ws
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    if not torch.cuda.is_current_stream_capturing():
        _scratch_cache[key] = ws
        while len(_scratch_cache) > _SCRATCH_MAX:
            _scratch_cache.popitem(last=False)
    return ws


def rmsnorm_bwd(g, x, w, eps, act=0, n_sum=1):
    """gymrl_rmsnorm_bwd / gymrl_rmsnorm_sum_bwd: (dL/dx [B, D], dL/dw [D]) of y = rmsnorm(x, w, eps, act=act); D <= 512.
    n_sum > 1: x is [B, n_sum, D] and the norm is of the blocks' sum — dL/dx [B, D] is every block's gradient."""
    B, D = x.shape[0], w.numel()
    ws = _scratch("rmsnorm_bwd", D, lib().gymrl_rmsnorm_bwd_workspace_bytes(C.c_int(D)), x.device)
    d_x, d_w = torch.empty(B, D, device=x.device), torch.empty_like(w)
    check(lib().gymrl_rmsnorm_sum_bwd(_ptr(g, torch.float32), _ptr(x, torch.float32), _ptr(w, torch.float32), C.c_int(B), C.c_int(D),
                                      C.c_int(n_sum), C.c_float(eps), C.c_int(act), _ptr(d_x), _ptr(d_w), _ptr(ws), _stream()),
          "gymrl_rmsnorm_sum_bwd")
    return d_x, d_w


def norm_proj_ok(D, n_out):
    """Shapes gymrl_norm_proj_fwd / _bwd take."""
    return 0 < D <= 256 and 0 < n_out <= 8


def _al16(t):
    """A tensor whose storage starts on a 16-byte boundary (a copy if the view does not): the D = 256 head-tail kernels are
    chosen by shape alone and require it (-22 otherwise), so that a result's bits never depend on where a view starts."""
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def norm_proj_fwd(x, norm_w, eps, W2, b2):
    """gymrl_norm_proj_fwd: RMSNorm(SiLU(x)) W2^T + b2 -> [B, n_out] in one launch."""
    B, D = x.shape
    if D == 256:
        x, norm_w, W2 = _al16(x), _al16(norm_w), _al16(W2)
    out = torch.empty(B, W2.shape[0], device=x.device)
    f = torch.float32
    check(lib().gymrl_norm_proj_fwd(_ptr(x, f), _ptr(norm_w, f), _ptr(W2, f), _ptr(b2, f, True), C.c_int(B), C.c_int(D),
                                    C.c_int(W2.shape[0]), C.c_float(eps), _ptr(out), _stream()), "gymrl_norm_proj_fwd")
    return out


def norm_proj_bwd(d_out, x, norm_w, eps, W2):
    """gymrl_norm_proj_bwd -> (d_x, d_norm_w, d_W2, d_b2)."""
    B, D = x.shape
    n_out = W2.shape[0]
    if D == 256:
        x, norm_w, W2 = _al16(x), _al16(norm_w), _al16(W2)
    ws = _scratch("norm_proj_bwd", (D, n_out), lib().gymrl_norm_proj_bwd_workspace_bytes(C.c_int(D), C.c_int(n_out)), x.device)
    d_x, d_nw, d_W2 = torch.empty_like(x), torch.empty_like(norm_w), torch.empty_like(W2)
    d_b2 = torch.empty(n_out, device=x.device)
    f = torch.float32
    check(lib().gymrl_norm_proj_bwd(_ptr(d_out, f), _ptr(x, f), _ptr(norm_w, f), _ptr(W2, f), C.c_int(B), C.c_int(D), C.c_int(n_out),
                                    C.c_float(eps), _ptr(d_x), _ptr(d_nw), _ptr(d_W2), _ptr(d_b2), _ptr(ws), _stream()),
          "gymrl_norm_proj_bwd")
    return d_x, d_nw, d_W2, d_b2


def mhc_sub_forward(h, norm_w, w, alpha, beta, lin_w, lin_b, sk_it):
    """gymrl_mhc_sub_forward: one hyper-connection sub-block forward (n = 2, D = 128) in one launch ->
    (pre, post, mix, stats, read, z, h_out).  h [B, 2, D], or [B, D]: the same row for both branches."""
    B, n, D = (h.shape[0], 2, h.shape[1]) if h.dim() == 2 else h.shape
    dev = h.device
    pre, post, mix = torch.empty(B, n, device=dev), torch.empty(B, n, device=dev), torch.empty(B, n, n, device=dev)
    stats, read, z = torch.empty(B, n * n + 2 * n + 1, device=dev), torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
    h_out = torch.empty(B, n, D, device=dev)
    f = torch.float32
    check(lib().gymrl_mhc_sub_forward(_ptr(h, f), C.c_int(h.dim() == 2), _ptr(norm_w, f), _ptr(w, f), _ptr(alpha, f), _ptr(beta, f), _ptr(lin_w, f),
                                      _ptr(lin_b, f), C.c_int(B), C.c_int(n), C.c_int(D), C.c_int(sk_it), _ptr(pre), _ptr(post),
                                      _ptr(mix), _ptr(stats), _ptr(read), _ptr(z), _ptr(h_out), _stream()), "gymrl_mhc_sub_forward")
    return pre, post, mix, stats, read, z, h_out


def mhc_sub_backward(g, h, z, pre, post, mix, stats, norm_w, w, alpha, lin_w, sum_branches=False):
    """gymrl_mhc_sub_backward: one hyper-connection sub-block backward (n = 2, D = 128) in one launch ->
    (d_z, d_h, d_norm_w, d_w, d_alpha, d_beta).  g / h may be [B, 128] (the same row for both branches); sum_branches: d_h
    comes back [B, 128], the branches' gradients added."""
    B, D = z.shape
    dev = z.device
    f = torch.float32
    for t, nm in ((g, "g"), (h, "h")):
        if t.shape not in ((B, 2, D), (B, D)) or not t.is_contiguous():
            raise ValueError(f"{nm}: expected a contiguous [B, 2, D] or [B, D] tensor")
    ws = _scratch("mhc_gates_bwd", (2, D), lib().gymrl_mhc_gates_bwd_workspace_bytes(C.c_int(2), C.c_int(D)), dev)
    Bp = (B + 15) // 16 * 16                               # the kernel writes whole 16-row tiles
    d_z = torch.empty(Bp, D, device=dev)[:B]
    d_h = (torch.empty(Bp, D, device=dev) if sum_branches else torch.empty(Bp, 2, D, device=dev))[:B]
    d_nw, d_w = torch.empty_like(norm_w), torch.empty_like(w)
    d_alpha, d_beta = torch.empty(3, device=dev), torch.empty(w.shape[1], device=dev)
    check(lib().gymrl_mhc_sub_backward(_ptr(g, f), C.c_int(g.dim() == 2), _ptr(h, f), C.c_int(h.dim() == 2), _ptr(z, f),
                                       _ptr(pre, f), _ptr(post, f), _ptr(mix, f), _ptr(stats, f), _ptr(norm_w, f), _ptr(w, f),
                                       _ptr(alpha, f), _ptr(lin_w, f), C.c_int(B), C.c_int(2), C.c_int(D), _ptr(d_z), _ptr(d_h),
                                       C.c_int(bool(sum_branches)), _ptr(d_nw), _ptr(d_w), _ptr(d_alpha), _ptr(d_beta), _ptr(ws),
                                       _stream()), "gymrl_mhc_sub_backward")
    return d_z, d_h, d_nw, d_w, d_alpha, d_beta


def mhc_policy(desc, obs, logits_out=None, value_out=None):
    """gymrl_mhc_policy_forward: PPO-full's whole rollout forward in one launch.  desc: a filled _lib.MhcPolicy (its pointers
    must stay alive: they are the modules' parameters); obs [B, obs_dim] -> (logits [B, n_act], value [B])."""
    B = obs.shape[0]
    logits = torch.empty(B, desc.n_act, device=obs.device) if logits_out is None else logits_out
    value = torch.empty(B, device=obs.device) if value_out is None else value_out
    check(lib().gymrl_mhc_policy_forward(C.byref(desc), _ptr(obs, torch.float32), C.c_int(B), _ptr(logits, torch.float32),
                                         _ptr(value, torch.float32), _stream()), "gymrl_mhc_policy_forward")
    return logits, value


def mhc_policy_pack(desc, image=None):
    """gymrl_mhc_policy_pack: the image of desc's wide operands (the sub-blocks' Linears and gate weights, the heads' first
    Linears) in the order the one-launch forward's lanes read them; returns the buffer (desc.image is the caller's to set —
    and to refresh: the image does not follow the parameters)."""
    n = int(lib().gymrl_mhc_policy_image_floats(C.c_int(desc.n_sub)))
    if image is None or image.numel() != n:
        image = torch.empty(n, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    check(lib().gymrl_mhc_policy_pack(C.byref(desc), _ptr(image, torch.float32), _stream()), "gymrl_mhc_policy_pack")
    return image


def sinkhorn(A, sk_it):
    """gymrl_sinkhorn: A [B, n, n] -> (u [B, n], v [B, n])."""
    B, n, _ = A.shape
    u, v = torch.empty(B, n, device=A.device), torch.empty(B, n, device=A.device)
    check(lib().gymrl_sinkhorn(_ptr(A, torch.float32), C.c_int(B), C.c_int(n), C.c_int(sk_it), _ptr(u), _ptr(v), _stream()),
          "gymrl_sinkhorn")
    return u, v


def mhc_read_fwd(pre, h):
    B, n, D = h.shape
    read = torch.empty(B, D, device=h.device)
    check(lib().gymrl_mhc_read_fwd(_ptr(pre, torch.float32), _ptr(h, torch.float32), C.c_int(B), C.c_int(n), C.c_int(D), _ptr(read),
                                   _stream()), "gymrl_mhc_read_fwd")
    return read


def mhc_read_bwd(g, pre, h, want_dh=True):
    """(d_pre, d_h); want_dh=False: d_pre only (the d_h term goes through gymrl_mhc_gates_bwd's d_read)."""
    B, n, D = h.shape
    d_pre = torch.empty(B, n, device=h.device)
    d_h = torch.empty_like(h) if want_dh else None
    check(lib().gymrl_mhc_read_bwd(_ptr(g, torch.float32), _ptr(pre, torch.float32), _ptr(h, torch.float32), C.c_int(B), C.c_int(n),
                                   C.c_int(D), _ptr(d_pre), None if d_h is None else _ptr(d_h), C.c_int(0), _stream()),
          "gymrl_mhc_read_bwd")
    return d_pre, d_h


def mhc_combine_bwd(g, post, mix, out, h, act=0, want_dh=True):
    """(d_post, d_mix, d_out, d_h) of gymrl_mhc_combine with the same act (silu: `out` is the raw z and d_out is dL/dz);
    want_dh=False: d_h = None (left to gymrl_mhc_gates_bwd's g_out term)."""
    B, n, D = h.shape
    dev = h.device
    d_post, d_mix = torch.empty(B, n, device=dev), torch.empty(B, n, n, device=dev)
    d_out = torch.empty(B, D, device=dev)
    d_h = torch.empty_like(h) if want_dh else None
    check(lib().gymrl_mhc_combine_bwd(_ptr(g, torch.float32), _ptr(post, torch.float32), _ptr(mix, torch.float32),
                                      _ptr(out, torch.float32), _ptr(h, torch.float32), C.c_int(B), C.c_int(n), C.c_int(D),
                                      C.c_int(act), _ptr(d_post), _ptr(d_mix), _ptr(d_out), None if d_h is None else _ptr(d_h),
                                      _stream()), "gymrl_mhc_combine_bwd")
    return d_post, d_mix, d_out, d_h


def mhc_gates_bwd(h, norm_w, w, alpha, pre, post, mix, stats, d_pre, d_post, d_mix, d_read=None, g_out=None):
    """gymrl_mhc_gates_bwd -> (d_h, d_norm_w, d_w, d_alpha, d_beta); n = 2 branches, n * D in (256, 512).  d_read [B, D] /
    g_out [B, n, D]: the read's and the combine's gradient paths into h, folded into d_h in the same pass."""
    B, n, D = h.shape
    dev = h.device
    ws = _scratch("mhc_gates_bwd", (n, D), lib().gymrl_mhc_gates_bwd_workspace_bytes(C.c_int(n), C.c_int(D)), dev)
    d_h, d_nw, d_w = torch.empty_like(h), torch.empty_like(norm_w), torch.empty_like(w)
    d_alpha, d_beta = torch.empty(3, device=dev), torch.empty(w.shape[1], device=dev)
    opt = lambda t: None if t is None else _ptr(t, torch.float32)   # noqa: E731
    check(lib().gymrl_mhc_gates_bwd(_ptr(h, torch.float32), _ptr(norm_w, torch.float32), _ptr(w, torch.float32),
                                    _ptr(alpha, torch.float32), _ptr(pre, torch.float32), _ptr(post, torch.float32),
                                    _ptr(mix, torch.float32), _ptr(stats, torch.float32), _ptr(d_pre, torch.float32),
                                    _ptr(d_post, torch.float32), _ptr(d_mix, torch.float32), opt(d_read), opt(g_out),
                                    C.c_int(B), C.c_int(n), C.c_int(D), _ptr(d_h), _ptr(d_nw), _ptr(d_w), _ptr(d_alpha),
                                    _ptr(d_beta), _ptr(ws), _stream()), "gymrl_mhc_gates_bwd")
    return d_h, d_nw, d_w, d_alpha, d_beta


# ------------------------------------------------- fused SAC vector step ---
def _addr(t):
    return None if t is None else _ptr(t).value


FUSED_MAX_BATCH = 8192     # row-slab kernels: B <= 256 as resident 2-D grids, above that slab-adjacent 1-D grids (offpolicy_step.hip slab_grid)


def sac_fused_shape_ok(B, D, A, H):
    """Shapes gymrl_sac_act_step / gymrl_sac_update take (include/gymrl.h): everything else runs layer by layer."""
    return 0 < B <= FUSED_MAX_BATCH and 0 < D <= 8 and 0 < A <= 4 and 4 <= H <= 256 and H % 4 == 0


def sac_update_workspace(B, D, A, H, device):
    # zeroed: the hand-off flags between the row phases' workgroups and gymrl_sac_step's phase counters live in it (zero before
    # the first launch, left zero)
    return torch.zeros(int(lib().gymrl_sac_update_workspace_bytes(C.c_int(B), C.c_int(D), C.c_int(A), C.c_int(H))),
                       dtype=torch.uint8, device=device)


def _sac_actor_params(dst, actor):
    for k, layer in enumerate((actor.fc1, actor.fc2, actor.mean, actor.log_std)):
        dst.w[k], dst.b[k] = _addr(layer.weight), _addr(layer.bias)


def _sac_critic_params(dst, critic):
    for k, layer in enumerate((critic.fc1, critic.fc2, critic.fc3, critic.fc4, critic.fc5, critic.fc6)):
        dst.w[k], dst.b[k] = _addr(layer.weight), _addr(layer.bias)


def sac_act_args(env, actor, ring, cap, bound, log_std_min, log_std_max, images=None):
    """A gymrl_sac_act_args with everything that does not change from step to step filled in (env, actor parameters —
    views of a flat buffer the optimiser updates in place — and the replay ring)."""
    a = SacActArgs()
    a.N, a.D, a.A, a.H = env.n, env.obs_dim, env.act_dim, actor.fc1.weight.shape[0]
    a.env_kind, a.env_state, a.env_seed, a.env_id0 = env.kind, _addr(env.state), env.seed, env.env_id0
    a.bound, a.log_std_min, a.log_std_max = float(bound), float(log_std_min), float(log_std_max)
    _sac_actor_params(a.actor, actor)
    a.r_state, a.r_action, a.r_reward, a.r_next, a.r_flag = (_addr(t) for t in ring)
    a.cap = cap
    a.images = _addr(images)
    return a


def sac_act_step(a, env, obs, obs_out, cursor=0, cursor_dev=None, eps=None, noise_seed=0, noise_counter=0, noise_counter_dev=None,
                 action_out=None, rew_out=None, done_out=None, ep_ret_out=None, ep_stats=None, launch=True):
    """gymrl_sac_act_step: Actor forward on obs [N, D], reparameterised draw, env step with auto-reset, replay rows at
    (cursor + env) % cap — ONE launch (sac_pendulum.py:278-283).  launch=False: the arguments are filled in only (sac_step
    launches them together with the update's)."""
    a.env_seed = env.seed                              # reset(seed=...) may have moved it
    a.obs, a.obs_out, a.eps = _ptr(obs, torch.float32).value, _ptr(obs_out, torch.float32).value, _addr(eps)
    a.noise_seed, a.noise_counter, a.noise_counter_dev = noise_seed, noise_counter, _addr(noise_counter_dev)
    a.cursor, a.cursor_dev = cursor, _addr(cursor_dev)
    a.action_out, a.rew_out, a.done_out, a.ep_ret_out, a.ep_stats = (_addr(t) for t in (action_out, rew_out, done_out, ep_ret_out, ep_stats))
    if launch:
        check(lib().gymrl_sac_act_step(C.byref(a), _stream()), "gymrl_sac_act_step")


def sac_images(H, device):
    """The eight weight images of the H x H layers (gymrl_sac_update_args.images), or None when H % 16 != 0."""
    return torch.zeros(8 * H * H, device=device) if H % 16 == 0 else None


def sac_pack_images(a):
    """gymrl_sac_pack_images: rebuild every image from the parameters as they are now."""
    check(lib().gymrl_sac_pack_images(C.byref(a), _stream()), "gymrl_sac_pack_images")


def sac_update_args(B, D, A, actor, critic, target, actor_opt, critic_opt, ring, cfg_scalars, log_alpha, alpha_m, alpha_v, sums,
                    alpha_loss, workspace, images=None):
    """A gymrl_sac_update_args with the per-trainer constants filled in.  cfg_scalars = (gamma, tau, bound, log_std_min,
    log_std_max, target_entropy, lr_alpha); actor_opt / critic_opt: FusedAdam over the modules' flat buffers."""
    a = SacUpdateArgs()
    a.B, a.D, a.A, a.H = B, D, A, actor.fc1.weight.shape[0]
    gamma, tau, bound, lo, hi, tent, lr_alpha = cfg_scalars
    a.gamma, a.tau, a.bound, a.log_std_min, a.log_std_max, a.target_entropy = float(gamma), float(tau), float(bound), float(lo), float(hi), float(tent)
    a.r_state, a.r_action, a.r_reward, a.r_next, a.r_flag = (_addr(t) for t in ring)
    _sac_actor_params(a.actor, actor)
    _sac_critic_params(a.critic, critic)
    _sac_critic_params(a.target, target)
    a.actor_p, a.actor_m, a.actor_v = _addr(actor_opt.p), _addr(actor_opt.m), _addr(actor_opt.v)
    a.critic_p, a.critic_m, a.critic_v = _addr(critic_opt.p), _addr(critic_opt.m), _addr(critic_opt.v)
    g = critic_opt.param_groups[0]
    a.beta1, a.beta2, a.eps_adam = g["betas"][0], g["betas"][1], g["eps"]
    a.log_alpha, a.alpha_m, a.alpha_v, a.lr_alpha = _addr(log_alpha), _addr(alpha_m), _addr(alpha_v), float(lr_alpha)
    a.sums, a.alpha_loss, a.workspace = _addr(sums), _addr(alpha_loss), _addr(workspace)
    a.images = _addr(images)
    return a


def sac_update(a, idx=None, idx_seed=0, idx_counter=0, idx_size=0, idx_dev=None, eps_next=None, eps_cur=None, noise_seed=0,
               noise_counter=0, noise_counter_dev=None, adam_critic=None, adam_actor=None, adam_critic_dev=None, adam_actor_dev=None,
               alpha_bias=(1.0, 1.0), alpha_bias_dev=None, launch=True):
    """gymrl_sac_update: SACTrainer.update() (sac_pendulum.py:213-267) as four launches.  adam_critic / adam_actor: the
    16-byte blocks of adam_bias() (host) or device views of them; alpha_bias = (1 - 0.9^t, 1 - 0.999^t).  launch=False: the
    arguments are filled in only (sac_step)."""
    a.idx, a.idx_seed, a.idx_counter, a.idx_size, a.idx_dev = _addr(idx), idx_seed, idx_counter, idx_size, _addr(idx_dev)
    a.eps_next, a.eps_cur = _addr(eps_next), _addr(eps_cur)
    a.noise_seed, a.noise_counter, a.noise_counter_dev = noise_seed, noise_counter, _addr(noise_counter_dev)
    for dst, blk in ((a.adam_critic, adam_critic), (a.adam_actor, adam_actor)):
        if blk is not None:
            vals = (C.c_float * 4).from_buffer_copy(blk)
            for k in range(4):
                dst[k] = vals[k]
    a.adam_critic_dev, a.adam_actor_dev = _addr(adam_critic_dev), _addr(adam_actor_dev)
    a.alpha_bias[0], a.alpha_bias[1], a.alpha_bias_dev = alpha_bias[0], alpha_bias[1], _addr(alpha_bias_dev)
    if launch:
        check(lib().gymrl_sac_update(C.byref(a), _stream()), "gymrl_sac_update")


def sac_step(act, upd):
    """gymrl_sac_step: the acting step and the update whose arguments sac_act_step(..., launch=False) and
    sac_update(..., launch=False) have filled in, as ONE launch."""
    check(lib().gymrl_sac_step(C.byref(act), C.byref(upd), _stream()), "gymrl_sac_step")


# --------------------------------------------- fused Rainbow vector step ---
def rainbow_fused_shape_ok(B, D, A, H):
    return 0 < B <= FUSED_MAX_BATCH and 0 < D <= 8 and 0 < A <= 3 and 4 <= H <= 256 and H % 4 == 0


def rainbow_update_workspace(B, D, A, H, device):
    # zeroed: the hand-off flags between gymrl_rainbow_update's three workgroups per slab live in it (zero before the first launch, left zero)
    return torch.zeros(int(lib().gymrl_rainbow_update_workspace_bytes(C.c_int(B), C.c_int(D), C.c_int(A), C.c_int(H))),
                       dtype=torch.uint8, device=device)


def rainbow_act_args(env, net, win, ring, cap, n_steps, gamma, max_episode_steps):
    """A gymrl_rainbow_act_args with the per-trainer constants filled in (net: DuelingNoisyNetwork — fc1 / fc2 are read in
    place, the noisy heads arrive per step as gymrl_noisy_combine's stacked output)."""
    a = RainbowActArgs()
    a.N, a.D, a.A, a.H = env.n, env.obs_dim, env.act_dim, net.fc1.weight.shape[0]
    a.env_kind, a.env_state, a.env_id0 = env.kind, _addr(env.state), env.env_id0
    a.fc1_w, a.fc1_b, a.fc2_w, a.fc2_b = _addr(net.fc1.weight), _addr(net.fc1.bias), _addr(net.fc2.weight), _addr(net.fc2.bias)
    a.max_episode_steps = int(max_episode_steps)
    a.w_state, a.w_action, a.w_reward, a.w_next, a.w_terminal, a.w_done = (_addr(t) for t in win)
    a.n_steps, a.gamma = int(n_steps), float(gamma)
    a.r_state, a.r_action, a.r_reward, a.r_next, a.r_flag = (_addr(t) for t in ring)
    a.cap = cap
    return a


def rainbow_act_step(a, env, obs, obs_out, head_w, head_b, pushes=0, cursor=0, push_dev=None, action_out=None, rew_out=None,
                     done_out=None, ep_ret_out=None, ep_stats=None, fc2_img=None):
    """gymrl_rainbow_act_step: greedy acting on the noisy Q + CartPole step + n-step push, one launch.  Returns what
    gymrl_nstep_push returns: whether rows were emitted (by the HOST's push count).  fc2_img: the forward weight image of fc2
    (noisy_combine(images=...)) — the caller vouches that it equals fc2.weight; None: read in place."""
    a.env_seed = env.seed
    a.fc2_img = _addr(fc2_img)
    a.obs, a.obs_out = _ptr(obs, torch.float32).value, _ptr(obs_out, torch.float32).value
    a.head_w, a.head_b = _ptr(head_w, torch.float32).value, _ptr(head_b, torch.float32).value
    a.pushes, a.cursor, a.push_dev = pushes, cursor, _addr(push_dev)
    a.action_out, a.rew_out, a.done_out, a.ep_ret_out, a.ep_stats = (_addr(t) for t in (action_out, rew_out, done_out, ep_ret_out, ep_stats))
    check(lib().gymrl_rainbow_act_step(C.byref(a), _stream()), "gymrl_rainbow_act_step")
    return pushes + 1 >= a.n_steps


def rainbow_update_args(B, D, A, policy, target, ring, gamma_n, loss_sum, d_head_w, d_head_b, workspace):
    a = RainbowUpdateArgs()
    a.B, a.D, a.A, a.H, a.gamma_n = B, D, A, policy.fc1.weight.shape[0], float(gamma_n)
    a.r_state, a.r_action, a.r_reward, a.r_next, a.r_flag = (_addr(t) for t in ring)
    a.p_fc1_w, a.p_fc1_b, a.p_fc2_w, a.p_fc2_b = (_addr(t) for t in (policy.fc1.weight, policy.fc1.bias, policy.fc2.weight, policy.fc2.bias))
    a.t_fc1_w, a.t_fc1_b, a.t_fc2_w, a.t_fc2_b = (_addr(t) for t in (target.fc1.weight, target.fc1.bias, target.fc2.weight, target.fc2.bias))
    a.d_fc1_w, a.d_fc1_b, a.d_fc2_w, a.d_fc2_b = (_addr(t) for t in (policy.fc1.weight.grad, policy.fc1.bias.grad, policy.fc2.weight.grad,
                                                                     policy.fc2.bias.grad))
    a.d_head_w, a.d_head_b, a.loss_sum, a.workspace = _addr(d_head_w), _addr(d_head_b), _addr(loss_sum), _addr(workspace)
    return a


def rainbow_update(a, idx, is_weight, head_w, head_b, td_out, split=None, phase=0, images=None):
    """gymrl_rainbow_update: gather + the three forwards + TD loss gradient + backward chain (rows), every weight gradient
    (tiles) — two launches.  head_w [3 (A+1), H] / head_b [3 (A+1)]: gymrl_noisy_combine's stacked output.
    images: (policy fc2 forward, policy fc2 input-gradient, target fc2 forward) weight images the caller vouches for, or None."""
    a.idx, a.is_weight = _ptr(idx, torch.int32).value, _addr(is_weight)
    a.p_fc2_img_f, a.p_fc2_img_b, a.t_fc2_img_f = (None, None, None) if images is None else (_addr(t) for t in images)
    a.head_w, a.head_b, a.td_out = _ptr(head_w, torch.float32).value, _ptr(head_b, torch.float32).value, _ptr(td_out, torch.float32).value
    # split: [(dw_mu, dw_sigma, db_mu, db_sigma, w_eps, b_eps)] of the advantage and the value layer -> gymrl_noisy_split's work
    # happens in the weight-gradient launch (None: the stacked gradient goes to d_head_w / d_head_b)
    a.split_heads = 0 if split is None else 1
    if split is not None:
        for l, (wm, wsg, bm, bsg, we, be) in enumerate(split):
            a.dw_mu[l], a.dw_sigma[l], a.db_mu[l], a.db_sigma[l] = _addr(wm), _addr(wsg), _addr(bm), _addr(bsg)
            a.w_eps[l], a.b_eps[l] = _addr(we), _addr(be)
    check(lib().gymrl_rainbow_update(C.byref(a), C.c_int(phase), _stream()), "gymrl_rainbow_update")

