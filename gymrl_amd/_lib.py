"""Loader for libgymrl_hip.so (the C-ABI of include/gymrl.h).

The library is built in-tree by gymrl_amd/csrc/Makefile (hipcc --offload-arch=gfx950)
and lives next to this file so that it travels with the repo snapshot.  Loading
never silently degrades: a missing library raises at first use.
"""
import ctypes as C
import os
import subprocess

# PyTorch-ROCm owns device memory and streams; importing it FIRST makes the process use
# one HIP runtime (torch's bundled libamdhip64) for both torch and this library.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GYMRL_HIP_LIB") or os.path.join(_HERE, "libgymrl_hip.so")   # override: A/B builds
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 3      # == GYMRL_ABI_VERSION of the include/gymrl.h this front-end was written against

_lib = None

# every symbol include/gymrl.h declares (tests/test_abi.py checks the export table)
SYMBOLS = [
    "gymrl_abi_version", "gymrl_device_ok",
    "gymrl_env_obs_dim", "gymrl_env_act_dim", "gymrl_env_is_discrete", "gymrl_env_max_steps",
    "gymrl_env_state_bytes", "gymrl_env_reset", "gymrl_env_step", "gymrl_env_refill", "gymrl_env_abandon",
    "gymrl_categorical_sample",
    "gymrl_gae_workspace_bytes", "gymrl_gae", "gymrl_gae_online_flush", "gymrl_gae_chunk", "gymrl_gae_dw", "gymrl_gae_decoupled", "gymrl_gae_decoupled_workspace_bytes",
    "gymrl_reduce_workspace_bytes", "gymrl_moments", "gymrl_normalize",
    "gymrl_ppo_loss_fwd_bwd", "gymrl_ppo_full_loss_fwd_bwd", "gymrl_ppo_rnn_loss_fwd_bwd",
    "gymrl_gru_cell_fwd", "gymrl_gru_cell_bwd", "gymrl_rnd_reward", "gymrl_permutation",
    "gymrl_pack_rollout", "gymrl_gather_minibatch", "gymrl_gather_rows", "gymrl_loss_blocks", "gymrl_reduce_rows",
    "gymrl_sqnorm", "gymrl_adam_step", "gymrl_clip_adam_step", "gymrl_adam_bias", "gymrl_store_scalars", "gymrl_soft_update",
    "gymrl_replay_append", "gymrl_replay_gather", "gymrl_uniform_indices", "gymrl_nstep_push",
    "gymrl_per_workspace_bytes", "gymrl_per_update", "gymrl_per_max_leaf", "gymrl_per_priorities", "gymrl_per_update_td",
    "gymrl_per_sample", "gymrl_noisy_noise", "gymrl_epsilon_greedy", "gymrl_dqn_td_loss",
    "gymrl_sac_sample_fwd", "gymrl_sac_sample_bwd", "gymrl_sac_target", "gymrl_sac_critic_loss",
    "gymrl_sac_actor_loss", "gymrl_sac_alpha_step", "gymrl_running_norm", "gymrl_reward_scaling",
    "gymrl_noisy_action", "gymrl_mse_loss", "gymrl_neg_mean_loss",
    "gymrl_dsac_target", "gymrl_dsac_critic_loss", "gymrl_dsac_actor_loss", "gymrl_dsac_alpha_step",
    "gymrl_mlp_packed_floats", "gymrl_mlp_pack", "gymrl_mlp_forward",
    "gymrl_mlp_train_workspace_bytes", "gymrl_linear_tanh_smallk", "gymrl_linear_smallk", "gymrl_tanh_inplace", "gymrl_tanh_bwd_colsum",
    "gymrl_linear_smallk_bwd", "gymrl_heads_fwd_tanh", "gymrl_heads_bwd", "gymrl_rollout_lunar", "gymrl_rollout_cartpole",
    "gymrl_gemm_workspace_bytes", "gymrl_linear_fwd", "gymrl_linear_bwd_input", "gymrl_linear_bwd_input_add",
    "gymrl_linear_bwd_weight_geometry", "gymrl_linear_bwd_weight",
    "gymrl_heads_loss_blocks", "gymrl_heads_loss_fwd_bwd", "gymrl_update_finalize",
    "gymrl_lin_workspace_bytes", "gymrl_lin_fwd", "gymrl_lin_bwd_input", "gymrl_lin_bwd_weight",
    "gymrl_noisy_combine", "gymrl_noisy_combine_images", "gymrl_noisy_split", "gymrl_dueling_bwd",
    "gymrl_mhc_gates", "gymrl_mhc_combine", "gymrl_rmsnorm", "gymrl_sinkhorn",
    "gymrl_mhc_read_fwd", "gymrl_mhc_read_bwd", "gymrl_mhc_combine_bwd",
    "gymrl_mhc_gates_bwd_workspace_bytes", "gymrl_mhc_gates_bwd", "gymrl_rmsnorm_bwd_workspace_bytes", "gymrl_rmsnorm_bwd", "gymrl_rmsnorm_sum_bwd", "gymrl_norm_proj_fwd", "gymrl_norm_proj_bwd_workspace_bytes", "gymrl_norm_proj_bwd",
    "gymrl_mhc_policy_forward", "gymrl_mhc_policy_image_floats", "gymrl_mhc_policy_pack", "gymrl_mhc_sub_forward", "gymrl_mhc_sub_backward", "gymrl_rollout_lunar_mhc",
    "gymrl_sac_update_workspace_bytes", "gymrl_sac_args_bytes", "gymrl_sac_act_step", "gymrl_sac_update", "gymrl_sac_step", "gymrl_sac_pack_images",
    "gymrl_rainbow_update_workspace_bytes", "gymrl_rainbow_args_bytes", "gymrl_rainbow_act_step", "gymrl_rainbow_update",
]


def build(force=False):
    """Compile the HIP library in-tree (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


class LinItem(C.Structure):
    """gymrl_lin_item (include/gymrl.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("x", "x2", "w", "b", "y", "dy", "dx", "dx2", "dw", "db")] +
                [("act", C.c_int), ("lo", C.c_float), ("hi", C.c_float), ("argmax", C.c_void_p)])


class NoisyLayer(C.Structure):
    """gymrl_noisy_layer (include/gymrl.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("w_mu", "w_sigma", "w_eps", "b_mu", "b_sigma", "b_eps", "w_eps_copy", "b_eps_copy",
                                          "dw_mu", "dw_sigma", "db_mu", "db_sigma")] +
                [("seed", C.c_uint64), ("counter", C.c_uint64), ("counter_dev", C.c_void_p), ("draw", C.c_int),
                 ("eval", C.c_int), ("n_out", C.c_int)])


class MhcSub(C.Structure):
    """gymrl_mhc_sub (include/gymrl.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("norm_w", "w", "alpha", "beta", "lin_w", "lin_b")]


class MhcHead(C.Structure):
    """gymrl_mhc_head (include/gymrl.h)."""
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("norm_w", C.c_void_p), ("norm_eps", C.c_float), ("w2", C.c_void_p),
                ("b2", C.c_void_p)]


class MhcPolicy(C.Structure):
    """gymrl_mhc_policy (include/gymrl.h)."""
    _fields_ = [("obs_dim", C.c_int), ("n_sub", C.c_int), ("n_act", C.c_int), ("sk_it", C.c_int), ("in_w", C.c_void_p),
                ("in_b", C.c_void_p), ("sub", MhcSub * 8), ("final_norm_w", C.c_void_p), ("final_norm_eps", C.c_float),
                ("head", MhcHead * 2), ("image", C.c_void_p)]


class PPOCfg(C.Structure):
    _fields_ = [("clip_eps", C.c_float), ("dual_clip", C.c_float), ("value_coef", C.c_float),
                ("entropy_coef", C.c_float)]


class GaeOnline(C.Structure):
    _fields_ = [("rew_prev", C.c_void_p), ("done_prev", C.c_void_p), ("val_prev", C.c_void_p),
                ("running", C.c_void_p), ("gae_workspace", C.c_void_p), ("t_prev", C.c_int), ("T", C.c_int),
                ("gamma", C.c_double), ("lam", C.c_double), ("lam2", C.c_double), ("running2", C.c_void_p)]


MLP_MAX_STAGES, MLP_MAX_WIDTH, MLP_MAX_INPUT = 8, 256, 64
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2


class MlpStage(C.Structure):
    _fields_ = [("W", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("in_dim", C.c_int),
                ("out_dim", C.c_int), ("act", C.c_int), ("src", C.c_int), ("dst", C.c_int), ("out_stride", C.c_int)]


class MlpDesc(C.Structure):
    _fields_ = [("n_stages", C.c_int), ("stage", MlpStage * MLP_MAX_STAGES)]


class RolloutLunarArgs(C.Structure):
    _fields_ = [("env_state", C.c_void_p), ("n_envs", C.c_int), ("seed", C.c_uint64), ("env_id0", C.c_int64),
                ("counter0", C.c_uint64), ("obs", C.c_void_p), ("act", C.c_void_p), ("logp", C.c_void_p),
                ("val", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p), ("ep_ret", C.c_void_p),
                ("next_value", C.c_void_p), ("noise_exp", C.c_void_p), ("gae_running", C.c_void_p),
                ("gae_workspace", C.c_void_p), ("gamma", C.c_double), ("lam", C.c_double), ("ep_stats", C.c_void_p),
                ("wg_ticks", C.c_void_p), ("T", C.c_int), ("t0", C.c_int), ("nsteps", C.c_int),
                ("ent", C.c_void_p), ("lam2", C.c_double), ("gae_running2", C.c_void_p),   # gymrl_rollout_lunar_mhc only
                ("refill", C.c_int), ("gae_carry", C.c_int)]


class SacActorParams(C.Structure):        # gymrl_sac_actor_params: fc1, fc2, mean, log_std
    _fields_ = [("w", C.c_void_p * 4), ("b", C.c_void_p * 4)]


class SacCriticParams(C.Structure):       # gymrl_sac_critic_params: fc1..fc6
    _fields_ = [("w", C.c_void_p * 6), ("b", C.c_void_p * 6)]


class SacActArgs(C.Structure):            # gymrl_sac_act_args (include/gymrl.h), field for field
    _fields_ = [("N", C.c_int), ("D", C.c_int), ("A", C.c_int), ("H", C.c_int), ("env_kind", C.c_int),
                ("env_state", C.c_void_p), ("env_seed", C.c_uint64), ("env_id0", C.c_int64),
                ("obs", C.c_void_p), ("obs_out", C.c_void_p), ("eps", C.c_void_p),
                ("noise_seed", C.c_uint64), ("noise_counter", C.c_uint64), ("noise_counter_dev", C.c_void_p),
                ("bound", C.c_float), ("log_std_min", C.c_float), ("log_std_max", C.c_float),
                ("actor", SacActorParams),
                ("r_state", C.c_void_p), ("r_action", C.c_void_p), ("r_reward", C.c_void_p), ("r_next", C.c_void_p),
                ("r_flag", C.c_void_p), ("cap", C.c_int64), ("cursor", C.c_int64), ("cursor_dev", C.c_void_p),
                ("action_out", C.c_void_p), ("rew_out", C.c_void_p), ("done_out", C.c_void_p), ("ep_ret_out", C.c_void_p),
                ("ep_stats", C.c_void_p), ("images", C.c_void_p)]


class SacUpdateArgs(C.Structure):         # gymrl_sac_update_args (include/gymrl.h), field for field
    _fields_ = [("B", C.c_int), ("D", C.c_int), ("A", C.c_int), ("H", C.c_int),
                ("gamma", C.c_float), ("bound", C.c_float), ("log_std_min", C.c_float), ("log_std_max", C.c_float),
                ("target_entropy", C.c_float), ("tau", C.c_double),
                ("r_state", C.c_void_p), ("r_action", C.c_void_p), ("r_reward", C.c_void_p), ("r_next", C.c_void_p),
                ("r_flag", C.c_void_p),
                ("idx", C.c_void_p), ("idx_seed", C.c_uint64), ("idx_counter", C.c_uint64), ("idx_size", C.c_int64),
                ("idx_dev", C.c_void_p),
                ("eps_next", C.c_void_p), ("eps_cur", C.c_void_p),
                ("noise_seed", C.c_uint64), ("noise_counter", C.c_uint64), ("noise_counter_dev", C.c_void_p),
                ("actor", SacActorParams), ("critic", SacCriticParams), ("target", SacCriticParams),
                ("actor_p", C.c_void_p), ("actor_m", C.c_void_p), ("actor_v", C.c_void_p),
                ("critic_p", C.c_void_p), ("critic_m", C.c_void_p), ("critic_v", C.c_void_p),
                ("adam_critic", C.c_float * 4), ("adam_actor", C.c_float * 4),
                ("adam_critic_dev", C.c_void_p), ("adam_actor_dev", C.c_void_p),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps_adam", C.c_double),
                ("log_alpha", C.c_void_p), ("alpha_m", C.c_void_p), ("alpha_v", C.c_void_p), ("lr_alpha", C.c_double),
                ("alpha_bias", C.c_double * 2), ("alpha_bias_dev", C.c_void_p),
                ("sums", C.c_void_p), ("alpha_loss", C.c_void_p), ("workspace", C.c_void_p), ("images", C.c_void_p)]


class RainbowActArgs(C.Structure):        # gymrl_rainbow_act_args (include/gymrl.h), field for field
    _fields_ = [("N", C.c_int), ("D", C.c_int), ("A", C.c_int), ("H", C.c_int), ("env_kind", C.c_int),
                ("env_state", C.c_void_p), ("env_seed", C.c_uint64), ("env_id0", C.c_int64),
                ("obs", C.c_void_p), ("obs_out", C.c_void_p),
                ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("max_episode_steps", C.c_int),
                ("w_state", C.c_void_p), ("w_action", C.c_void_p), ("w_reward", C.c_void_p), ("w_next", C.c_void_p),
                ("w_terminal", C.c_void_p), ("w_done", C.c_void_p),
                ("n_steps", C.c_int), ("pushes", C.c_int64), ("gamma", C.c_double),
                ("r_state", C.c_void_p), ("r_action", C.c_void_p), ("r_reward", C.c_void_p), ("r_next", C.c_void_p),
                ("r_flag", C.c_void_p), ("cap", C.c_int64), ("cursor", C.c_int64), ("push_dev", C.c_void_p),
                ("action_out", C.c_void_p), ("rew_out", C.c_void_p), ("done_out", C.c_void_p), ("ep_ret_out", C.c_void_p),
                ("ep_stats", C.c_void_p), ("fc2_img", C.c_void_p)]


class RainbowUpdateArgs(C.Structure):     # gymrl_rainbow_update_args (include/gymrl.h), field for field
    _fields_ = [("B", C.c_int), ("D", C.c_int), ("A", C.c_int), ("H", C.c_int), ("gamma_n", C.c_float),
                ("r_state", C.c_void_p), ("r_action", C.c_void_p), ("r_reward", C.c_void_p), ("r_next", C.c_void_p),
                ("r_flag", C.c_void_p), ("idx", C.c_void_p), ("is_weight", C.c_void_p),
                ("p_fc1_w", C.c_void_p), ("p_fc1_b", C.c_void_p), ("p_fc2_w", C.c_void_p), ("p_fc2_b", C.c_void_p),
                ("t_fc1_w", C.c_void_p), ("t_fc1_b", C.c_void_p), ("t_fc2_w", C.c_void_p), ("t_fc2_b", C.c_void_p),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("td_out", C.c_void_p), ("loss_sum", C.c_void_p),
                ("d_fc1_w", C.c_void_p), ("d_fc1_b", C.c_void_p), ("d_fc2_w", C.c_void_p), ("d_fc2_b", C.c_void_p),
                ("d_head_w", C.c_void_p), ("d_head_b", C.c_void_p), ("workspace", C.c_void_p),
                ("split_heads", C.c_int), ("dw_mu", C.c_void_p * 2), ("dw_sigma", C.c_void_p * 2), ("db_mu", C.c_void_p * 2),
                ("db_sigma", C.c_void_p * 2), ("w_eps", C.c_void_p * 2), ("b_eps", C.c_void_p * 2),
                ("p_fc2_img_f", C.c_void_p), ("p_fc2_img_b", C.c_void_p), ("t_fc2_img_f", C.c_void_p)]


class WeightImage(C.Structure):           # gymrl_weight_image
    _fields_ = [("W", C.c_void_p), ("H", C.c_int), ("img_fwd", C.c_void_p), ("img_bwd", C.c_void_p)]


class PPOFullCfg(C.Structure):
    _fields_ = [("clip_eps_min", C.c_float), ("clip_eps_max", C.c_float), ("dual_clip", C.c_float),
                ("erc_beta_low", C.c_float), ("erc_beta_high", C.c_float), ("entropy_coef", C.c_float),
                ("entropy_coef_dev", C.c_void_p)]


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
                "gymrl_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.gymrl_abi_version.restype = C.c_int
        got = L.gymrl_abi_version()
        if got != ABI_VERSION:       # a stale .so would take mis-aligned arguments silently
            raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, this package needs {ABI_VERSION}: rebuild it "
                               f"with `make -C {CSRC}`")
        L.gymrl_env_state_bytes.restype = C.c_size_t
        L.gymrl_gae_workspace_bytes.restype = C.c_size_t
        L.gymrl_gae_decoupled_workspace_bytes.restype = C.c_size_t
        L.gymrl_reduce_workspace_bytes.restype = C.c_size_t
        L.gymrl_per_workspace_bytes.restype = C.c_size_t
        L.gymrl_mlp_packed_floats.restype = C.c_size_t
        L.gymrl_mhc_policy_image_floats.restype = C.c_size_t
        L.gymrl_mlp_train_workspace_bytes.restype = C.c_size_t
        L.gymrl_gemm_workspace_bytes.restype = C.c_size_t
        L.gymrl_lin_workspace_bytes.restype = C.c_size_t
        L.gymrl_mhc_gates_bwd_workspace_bytes.restype = C.c_size_t
        L.gymrl_rmsnorm_bwd_workspace_bytes.restype = C.c_size_t
        L.gymrl_norm_proj_bwd_workspace_bytes.restype = C.c_size_t
        L.gymrl_sac_update_workspace_bytes.restype = C.c_size_t
        L.gymrl_sac_args_bytes.restype = C.c_size_t
        L.gymrl_rainbow_update_workspace_bytes.restype = C.c_size_t
        L.gymrl_rainbow_args_bytes.restype = C.c_size_t
        if L.gymrl_rainbow_args_bytes(0) != C.sizeof(RainbowActArgs) or L.gymrl_rainbow_args_bytes(1) != C.sizeof(RainbowUpdateArgs):
            raise RuntimeError("gymrl_amd/_lib.py: RainbowActArgs / RainbowUpdateArgs do not mirror include/gymrl.h "
                               f"({C.sizeof(RainbowActArgs)} / {C.sizeof(RainbowUpdateArgs)} bytes here, "
                               f"{L.gymrl_rainbow_args_bytes(0)} / {L.gymrl_rainbow_args_bytes(1)} in the library)")
        if L.gymrl_sac_args_bytes(0) != C.sizeof(SacActArgs) or L.gymrl_sac_args_bytes(1) != C.sizeof(SacUpdateArgs):
            raise RuntimeError("gymrl_amd/_lib.py: SacActArgs / SacUpdateArgs do not mirror include/gymrl.h "
                               f"({C.sizeof(SacActArgs)} / {C.sizeof(SacUpdateArgs)} bytes here, "
                               f"{L.gymrl_sac_args_bytes(0)} / {L.gymrl_sac_args_bytes(1)} in the library)")
        for name in SYMBOLS:
            if name.endswith(("_bytes", "_floats")):
                continue
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def lib_sha256():
    """sha256 of the library file that lib() loads — the identity of the binary a measurement was taken on
    (profiles/*_pmc_summary.json carry it; bench.py refuses a counter summary taken on another binary)."""
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (EINVAL)" if rc == -22 else ""))
