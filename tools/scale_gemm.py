import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gymrl_amd import ops
from tools.micro_gemm import timeit
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Bmax = 1048576
x = torch.randn(Bmax, 256, device=dev, generator=g); dy2 = torch.randn(Bmax, 256, device=dev, generator=g)
W2 = torch.randn(256, 256, device=dev, generator=g) / 16; b2 = torch.randn(256, device=dev, generator=g)
y2 = torch.empty(Bmax, 256, device=dev); dW2 = torch.empty(256, 256, device=dev); ws = ops.gemm_workspace(dev)
for B in (8192, 32768, 65536, 131072, 262144, 524288, 1048576):
    f = 2.0 * B * 65536
    t1 = timeit(lambda: ops.linear_fwd(x[:B], W2, b2, y2[:B], act=True))
    t2 = timeit(lambda: ops.linear_bwd_weight(dy2[:B], x[:B], dW2, ws))
    t3 = timeit(lambda: torch.mm(x[:B], W2.t(), out=y2[:B]))
    print(f"B={B:8d} fwd {t1:8.1f} us {f/t1*1e-6:6.1f} TF | dW {t2:8.1f} us {f/t2*1e-6:6.1f} TF | lib mm {t3:8.1f} us {f/t3*1e-6:6.1f} TF", flush=True)
