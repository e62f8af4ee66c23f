"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/gymrl.h declares; no compute is launched (no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gymrl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"#ifdef GYMRL_PROF_BUILD.*?#endif", "", hdr, flags=re.S)      # probe-build-only declarations
    return sorted(set(re.findall(r"\b(gymrl_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from gymrl_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/gymrl.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "gymrl_amd/_lib.py SYMBOLS out of sync with include/gymrl.h"
    assert L.gymrl_abi_version() == 3 == _lib.ABI_VERSION
    # the product library carries no diagnostic switches (timing-only kernel variants live in the probe build)
    assert not hasattr(L, "gymrl_gemm_config")


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """_lib.lib() compares the library's ABI version with the one this front-end was written against."""
    from gymrl_amd import _lib
    import pytest
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", 999)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.lib()
    monkeypatch.setattr(_lib, "ABI_VERSION", 3)
    assert _lib.lib().gymrl_abi_version() == 3


def test_size_queries_need_no_gpu():
    from gymrl_amd import _lib
    L = _lib.lib()
    assert L.gymrl_env_obs_dim(0) == 4 and L.gymrl_env_obs_dim(1) == 3 and L.gymrl_env_obs_dim(2) == 8
    assert L.gymrl_env_act_dim(0) == 2 and L.gymrl_env_act_dim(2) == 4
    assert L.gymrl_env_max_steps(0) == 500 and L.gymrl_env_max_steps(1) == 200 and L.gymrl_env_max_steps(2) == 1000
    assert L.gymrl_env_state_bytes(0, 4096) > 0
    assert L.gymrl_gae_workspace_bytes(2048, 4096) > 0
    assert L.gymrl_env_obs_dim(99) == -22          # EINVAL, like the header says


def test_bad_arguments_return_einval_without_gpu():
    from gymrl_amd import _lib
    L = _lib.lib()
    null = ctypes.c_void_p(None)
    assert L.gymrl_gae(null, null, null, null, 4, 4, ctypes.c_double(0.99), ctypes.c_double(0.95), null, null,
                       null, 0, null, null) == -22
    assert L.gymrl_soft_update(null, null, ctypes.c_int64(8), ctypes.c_double(0.005), null) == -22


def test_new_entry_points_validate_arguments_without_gpu():
    """Argument validation happens before any HIP call: NULL pointers, unsupported shapes and
    inconsistent descriptors return -EINVAL (22) on a machine with no GPU at all."""
    from gymrl_amd import _lib
    L = _lib.lib()
    null, i64, dbl = ctypes.c_void_p(None), ctypes.c_int64, ctypes.c_double
    assert L.gymrl_mlp_packed_floats(4, 8) == 1 * 4 * 256 and L.gymrl_mlp_packed_floats(256, 256) == 16 * 16 * 256
    assert L.gymrl_mlp_packed_floats(0, 8) == 0
    assert L.gymrl_mlp_pack(null, 4, 8, null, null) == -22
    desc = _lib.MlpDesc()
    desc.n_stages = 0
    fake = ctypes.c_void_p(256)                       # never dereferenced: validation fails first
    assert L.gymrl_mlp_forward(fake, 16, 8, ctypes.byref(desc), null) == -22          # no stages
    desc.n_stages = 1
    desc.stage[0].W, desc.stage[0].in_dim, desc.stage[0].out_dim = 256, 8, 4
    desc.stage[0].src, desc.stage[0].dst, desc.stage[0].act = -1, 1, 7
    assert L.gymrl_mlp_forward(fake, 16, 8, ctypes.byref(desc), null) == -22          # unknown activation
    desc.stage[0].act, desc.stage[0].dst = 0, -1
    assert L.gymrl_mlp_forward(fake, 16, 8, ctypes.byref(desc), null) == -22          # dst == -1 without `out`
    assert L.gymrl_mlp_forward(fake, 16, 100, ctypes.byref(desc), null) == -22         # input wider than supported
    assert L.gymrl_mlp_train_workspace_bytes(256, 8, 4) >= 4 * 1024 * 9 * 256
    assert L.gymrl_linear_tanh_smallk(fake, fake, null, i64(8), 8, 48, fake, null) == -22     # C not a power of two
    assert L.gymrl_linear_tanh_smallk(fake, fake, null, i64(8), 5, 64, fake, null) == -22     # unsupported input width
    assert L.gymrl_tanh_inplace(fake, i64(6), null, 0, null) == -22                            # n % 4 != 0
    assert L.gymrl_heads_bwd(fake, fake, fake, i64(8), 64, 3, fake, fake, fake, fake, fake, fake, fake, fake, 0, null,
                             fake, null) == -22                                                # A not in {2, 4}
    assert L.gymrl_heads_fwd_tanh(null, i64(8), 64, 4, null, fake, null, fake, null, fake, fake, 1, null) == -22
    args = _lib.RolloutLunarArgs()
    assert L.gymrl_rollout_lunar(ctypes.byref(args), ctypes.byref(desc), null) == -22          # NULL slabs
    # the fused off-policy steps: NULL / empty argument blocks are refused before any launch
    act, upd = _lib.SacActArgs(), _lib.SacUpdateArgs()
    assert L.gymrl_sac_act_step(ctypes.byref(act), null) == -22 and L.gymrl_sac_update(ctypes.byref(upd), null) == -22
    assert L.gymrl_sac_step(ctypes.byref(act), ctypes.byref(upd), null) == -22
    assert L.gymrl_sac_step(null, ctypes.byref(upd), null) == -22 and L.gymrl_sac_step(ctypes.byref(act), null, null) == -22
    # empty work is a no-op that returns 0 without launching anything
    assert L.gymrl_tanh_inplace(fake, i64(0), null, 0, null) == 0
    assert L.gymrl_linear_tanh_smallk(fake, fake, null, i64(0), 8, 64, fake, null) == 0


def test_mhc_entry_points_validate_arguments_without_gpu():
    """PPO-full's mHC kernels: shapes outside what a kernel is written for are -EINVAL before any HIP call."""
    from gymrl_amd import _lib
    L = _lib.lib()
    null, f32 = ctypes.c_void_p(None), ctypes.c_float
    fake = ctypes.c_void_p(256)                       # never dereferenced: validation fails first
    gates = lambda n, D, stats: L.gymrl_mhc_gates(fake, fake, fake, fake, fake, 8, n, D, 10, fake, fake, fake, fake, stats, null)  # noqa: E731
    assert gates(3, 128, null) == -22                  # branches: 2 or 4
    assert gates(2, 6, null) == -22                    # D % 4
    assert gates(4, 64, fake) == -22                   # the read-out sums exist for n = 2, n * D in (256, 512) only
    assert gates(2, 64, fake) == -22
    assert L.gymrl_mhc_gates(fake, fake, fake, fake, fake, 0, 2, 128, 10, fake, fake, fake, fake, fake, null) == 0   # empty batch
    assert L.gymrl_mhc_combine(fake, fake, fake, fake, 8, 2, 128, 1, fake, null) == -22          # act: none or SiLU
    assert L.gymrl_mhc_combine_bwd(fake, fake, fake, fake, fake, 8, 2, 128, 2, fake, fake, fake, null, null) == -22
    assert L.gymrl_mhc_combine_bwd(fake, fake, fake, fake, fake, 0, 2, 128, 5, fake, fake, fake, null, null) == 0    # d_h may be NULL
    assert L.gymrl_mhc_read_bwd(fake, fake, fake, 8, 3, 128, fake, null, 0, null) == -22
    bwd = lambda n, D, stats, ws: L.gymrl_mhc_gates_bwd(fake, fake, fake, fake, fake, fake, fake, stats, fake, fake, fake, null, null,  # noqa: E731
                                                        8, n, D, fake, fake, fake, fake, fake, ws, null)
    assert bwd(4, 64, fake, fake) == -22 and bwd(2, 64, fake, fake) == -22 and bwd(2, 128, null, fake) == -22
    assert bwd(2, 128, fake, null) == -22              # workspace required
    assert L.gymrl_mhc_gates_bwd_workspace_bytes(2, 128) >= 512 * 2315 * 4
    assert L.gymrl_mhc_gates_bwd_workspace_bytes(2, 256) >= 2 * 512 * 2315 * 4
    assert L.gymrl_rmsnorm(fake, fake, 8, 128, 1, f32(1e-6), 3, fake, null) == -22
    assert L.gymrl_rmsnorm_bwd(fake, fake, fake, 8, 513, f32(1e-6), 0, fake, fake, fake, null) == -22      # D <= 512
    assert L.gymrl_rmsnorm_bwd(fake, fake, fake, 8, 128, f32(1e-6), 0, fake, fake, null, null) == -22      # workspace required
    assert L.gymrl_rmsnorm_bwd_workspace_bytes(256) >= 2048 * 256 * 4
    pol = _lib.MhcPolicy()
    assert L.gymrl_mhc_policy_forward(null, fake, 8, fake, fake, null) == -22
    # the packed operand image: its size is a function of the sub-block count; pack refuses NULL / misaligned / incomplete inputs
    assert L.gymrl_mhc_policy_image_floats(4) == 4 * (128 * 128 + 256 * 8) + 2 * 256 * 128 and L.gymrl_mhc_policy_image_floats(9) == 0
    assert L.gymrl_mhc_policy_pack(null, fake, null) == -22 and L.gymrl_mhc_policy_pack(ctypes.byref(pol), null, null) == -22
    assert L.gymrl_mhc_policy_pack(ctypes.byref(pol), ctypes.c_void_p(260), null) == -22        # image not 16-byte aligned
    assert L.gymrl_mhc_policy_pack(ctypes.byref(pol), fake, null) == -22                        # NULL parameters
    pol.obs_dim, pol.n_sub, pol.n_act, pol.sk_it = 8, 2, 4, 10
    assert L.gymrl_mhc_policy_forward(ctypes.byref(pol), fake, 8, fake, fake, null) == -22       # NULL parameters
    pol.in_w = pol.in_b = pol.final_norm_w = 256
    assert L.gymrl_mhc_policy_forward(ctypes.byref(pol), fake, 8, fake, fake, null) == -22       # NULL sub-block parameters
    pol.n_sub, pol.obs_dim = 9, 8
    assert L.gymrl_mhc_policy_forward(ctypes.byref(pol), fake, 8, fake, fake, null) == -22       # more than 8 sub-blocks
    pol.n_sub, pol.obs_dim = 0, 17
    assert L.gymrl_mhc_policy_forward(ctypes.byref(pol), fake, 8, fake, fake, null) == -22       # more than 16 observations
    # the weight gradient's slice count follows the shape (64 x 64 blocks from 16384 rows on): the workspace query says so
    small, big = L.gymrl_lin_workspace_bytes(8192, 128, 128, 1), L.gymrl_lin_workspace_bytes(262144, 128, 128, 1)
    assert small == 32 * (128 * 128 + 128) * 4 and big == 512 * (128 * 128 + 128) * 4


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from gymrl_amd import ops
    x = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gae(x, x, x.to(torch.uint8), x[0], 0.99, 0.95, variant=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gymrl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libgymrl_oracle", "oracle/", "orc_"):
                    assert needle not in src, f"{f} uses the oracle ({needle})"
