"""ctypes wrappers of the split-bf16 PROBE library (tools/probes/libgymrl_probe_sb.so, `make -C tools/probes`): the four entry
points that lived in libgymrl_hip.so until round 5.  Used by test_gemm_sb_gpu.py (this directory) and tools/{micro,abl,pmc}_*sb.py."""
import ctypes as C
import os

import torch

from gymrl_amd.ops import _ptr, _stream, check

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgymrl_probe_sb.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: make -C tools/probes")
        _LIB = C.CDLL(path)
        for name in ("gymrl_linear_fwd_sb", "gymrl_split_planes", "gymrl_linear_fwd_sb_planes", "gymrl_linear_bwd_input_sb"):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def linear_fwd_sb(x, W, b, out, act=True):
    """gymrl_linear_fwd_sb: linear_fwd's product on the bf16 matrix cores (three-way split operands, f32-accurate; opt-in)."""
    B, K = x.shape
    check(lib().gymrl_linear_fwd_sb(_ptr(x, torch.float32), _ptr(W, torch.float32), _ptr(b, torch.float32, True),
                                    C.c_int64(B), C.c_int(K), C.c_int(W.shape[0]), C.c_int(int(act)), _ptr(out, torch.float32),
                                    _stream()), "gymrl_linear_fwd_sb")
    return out


def split_planes(x, out=None):
    """gymrl_split_planes: the three bf16 planes [3, *x.shape] (as int16 storage) of a float32 tensor."""
    out = torch.empty((3,) + tuple(x.shape), dtype=torch.int16, device=x.device) if out is None else out
    check(lib().gymrl_split_planes(_ptr(x, torch.float32), C.c_int64(x.numel()), _ptr(out, torch.int16), _stream()), "gymrl_split_planes")
    return out


def linear_fwd_sb_planes(x_planes, W, b, out, act=True):
    """gymrl_linear_fwd_sb_planes: linear_fwd_sb on activations already split into planes [3, B, 256]."""
    _, B, K = x_planes.shape
    check(lib().gymrl_linear_fwd_sb_planes(_ptr(x_planes, torch.int16), _ptr(W, torch.float32), _ptr(b, torch.float32, True),
                                           C.c_int64(B), C.c_int(K), C.c_int(W.shape[0]), C.c_int(int(act)), _ptr(out, torch.float32),
                                           _stream()), "gymrl_linear_fwd_sb_planes")
    return out


def linear_bwd_input_sb(dy, W, H, dx):
    """gymrl_linear_bwd_input_sb: linear_bwd_input's product on the bf16 matrix cores (opt-in)."""
    B, N = dy.shape
    check(lib().gymrl_linear_bwd_input_sb(_ptr(dy, torch.float32), _ptr(W, torch.float32), _ptr(H, torch.float32, True),
                                          C.c_int64(B), C.c_int(N), C.c_int(W.shape[1]), _ptr(dx, torch.float32), _stream()),
          "gymrl_linear_bwd_input_sb")
    return dx
