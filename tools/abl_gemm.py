import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gymrl_amd import ops
from tools.micro_gemm import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B = 262144
x = torch.randn(B, 256, device=dev, generator=g); dy2 = torch.randn(B, 256, device=dev, generator=g)
dW2 = torch.empty(256, 256, device=dev); ws = ops.gemm_workspace(dev)
f = 2.0 * B * 65536
for abl in (0, 0, 1, 2):
    ops.gemm_config(4, abl)
    t = timeit(lambda: ops.linear_bwd_weight(dy2, x, dW2, ws))
    ff = f
    print(f"abl={abl} {t:8.1f} us {ff/t*1e-6:6.1f} TF", flush=True)
ops.gemm_config(4, 0)
W2 = torch.randn(256, 256, device=dev, generator=g) / 16; b2 = torch.randn(256, device=dev, generator=g); y2 = torch.empty(B, 256, device=dev)
for abl in (0, 0, 8, 4, 12):
    ops.gemm_config(5, abl)
    t = timeit(lambda: ops.linear_fwd(x, W2, b2, y2, act=True))
    print(f"ws abl={abl} {t:8.1f} us {f/t*1e-6:6.1f} TF", flush=True)
ops.gemm_config(5, 0)
