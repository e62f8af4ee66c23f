#!/usr/bin/env python3
"""Launch the GAE (blocked), PPO-loss and update-path kernels a few times at bench shapes for rocprofv3 --pmc
passes (FETCH_SIZE / WRITE_SIZE are collected in separate runs).  T x N = 2048 x 4096; the loss runs
once over the whole rollout (8.39 M samples) and at the bench's minibatch size (262144)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd import ops
dev = torch.device("cuda:0")
T, N = 2048, 4096
g = torch.Generator(device=dev).manual_seed(1)
rew, val = torch.randn(T, N, device=dev, generator=g), torch.randn(T, N, device=dev, generator=g)
done = (torch.rand(T, N, device=dev, generator=g) < 1 / 300).to(torch.uint8)
nv = torch.randn(N, device=dev, generator=g)
adv, ret = torch.empty_like(rew), torch.empty_like(rew)
ws, mom = ops.gae_workspace(T, N, dev), torch.zeros(3, dtype=torch.float64, device=dev)
B = T * N
logits, v = torch.randn(B, 4, device=dev, generator=g), torch.randn(B, device=dev, generator=g)
act = torch.randint(0, 4, (B,), device=dev, generator=g, dtype=torch.int32)
lpo = torch.randn(B, device=dev, generator=g) * 0.1 - 1.4
dl, dv = torch.empty_like(logits), torch.empty_like(v)
met = torch.zeros(5, dtype=torch.float64, device=dev)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)     # evict the 256 MiB Infinity Cache between launches
ops.gae(rew, val, done, nv, 0.99, 0.95, adv, ret, mom, 1, ws)     # primes the chunk maps (the rollout does this online)
for it in range(3):
    junk.zero_()
    ops.gae(rew, val, done, nv, 0.99, 0.95, adv, ret, mom, 2, ws)
    junk.zero_()
    ops.ppo_loss_fwd_bwd(logits, v, act, lpo, adv.view(-1), ret.view(-1), (0.2, 3.0, 0.5, 0.01), None, mom, dl, dv, met)
    junk.zero_()
    mb = 262144
    ops.ppo_loss_fwd_bwd(logits[:mb], v[:mb], act[:mb], lpo[:mb], adv.view(-1)[:mb], ret.view(-1)[:mb], (0.2, 3.0, 0.5, 0.01),
                         None, mom, dl[:mb], dv[:mb], met)
# update-path passes (csrc/mlp_train.hip) at the bench's minibatch: B = 262144 rows, C = 256, obs 8, A = 4
Bm, C, D, A = 262144, 256, 8, 4
x = torch.randn(Bm, D, device=dev, generator=g)
W1, b1 = torch.randn(C, D, device=dev, generator=g) * 0.3, torch.zeros(C, device=dev)
H1, dH = torch.empty(Bm, C, device=dev), torch.randn(Bm, C, device=dev, generator=g)
Hac, dZac = torch.tanh(torch.randn(Bm, 2 * C, device=dev, generator=g)), torch.empty(Bm, 2 * C, device=dev)
dlg, dvv = torch.randn(Bm, A, device=dev, generator=g), torch.randn(Bm, device=dev, generator=g)
Wa2, Wc2 = torch.randn(A, C, device=dev, generator=g), torch.randn(1, C, device=dev, generator=g)
gws = ops.mlp_train_workspace(C, D, A, dev)
dbac, dWa2, dba2, dWc2, dbc2 = (torch.empty(n, device=dev) for n in (2 * C, A * C, A, C, 1))
dW1, db1 = torch.empty(C, D, device=dev), torch.empty(C, device=dev)
for it in range(3):
    junk.zero_()
    ops.linear_tanh_smallk(x, W1, b1, H1)
    junk.zero_()
    ops.tanh_inplace(dZac.copy_(Hac))
    junk.zero_()
    ops.heads_fwd_tanh(dZac.copy_(Hac), Wa2, dba2, Wc2, dbc2, dlg, dvv, dbac, store_h=False)
    junk.zero_()
    ops.heads_bwd(Hac, dlg, dvv, Wa2, Wc2, dZac, dbac, dWa2.view(A, C), dba2, dWc2.view(1, C), dbc2, gws, pre_activation=True,
                  bac=dbac.clone())
    junk.zero_()
    ops.tanh_bwd_colsum(dH, H1, db1, gws)
    junk.zero_()
    ops.linear_smallk_bwd(dH, H1, x, dW1, db1, gws)
torch.cuda.synchronize()
print("done")
