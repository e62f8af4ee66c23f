#!/usr/bin/env python3
"""Which workgroups of the persistent rollout are the slow ones, and what state are their envs in?  Probe build
(`make -C gymrl_amd/csrc prof`).  Prints the section timers of the five slowest workgroups and the world words of the
slowest one's envs (bodies, joint impulses, contact counts, subnormal words)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GYMRL_HIP_LIB", os.path.join(ROOT, "gymrl_amd", "libgymrl_hip_prof.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gymrl_amd.ppo_lunarlander import Config, PPOTrainer  # noqa: E402

N, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = Config()
cfg.num_envs, cfg.update_freq, cfg.seed, cfg.rollout_chunk = N, T, 0, T
sys.stdout = open(os.devnull, "w")
tr = PPOTrainer(cfg)
sys.stdout = sys.__stdout__
tr.collect_rollout()
tr.rollout_count = 0
G = N // 16
tr._wg_ticks = torch.zeros(2 * G + 24 * G + 3 * N, dtype=torch.int64, device=tr.device)
tr.collect_rollout()
torch.cuda.synchronize()
tk = tr._wg_ticks.cpu().numpy()
busy = (tk[1:2 * G:2] - tk[0:2 * G:2]) / 100.0 / T
sec = tk[2 * G:26 * G].reshape(G, 24) / 100.0 / T
raw = tk[2 * G:26 * G].reshape(G, 24)
order = np.argsort(-busy)
print("wg   busy  engines  init  sweeps  (contact / none)  positions  pos-iters/step  forward")
for g in list(order[:8]) + list(order[G // 2:G // 2 + 3]):
    print(f"{g:4d} {busy[g]:6.1f} {sec[g,0]:7.1f} {sec[g,1]:5.1f} {sec[g,2]:7.1f}  ({sec[g,6]:6.1f} / {sec[g,7]:6.1f}) {sec[g,4]:9.1f} "
          f"{raw[g,11]/T:9.2f} {sec[g,8]:8.1f}")
st = tr.env.state.cpu()
nb = 144 * 4 * N                                  # the live worlds: 144 words per env, [word][env]
wi = st[:nb].view(torch.int32).view(144, N).numpy()
wf = st[:nb].view(torch.float32).view(144, N).numpy()
off = (nb + 255) // 256 * 256 + (8 * N + 255) // 256 * 256
ep_len = st[off:off + 4 * N].view(torch.int32).numpy()
g = int(order[0])
print(f"\nslowest workgroup {g}: envs {16*g}..{16*g+15}")
for e in range(16 * g, 16 * g + 16):
    f = wf[:, e]; u = wi[:, e]
    expo = (u.view(np.uint32) >> 23) & 0xff
    man = u.view(np.uint32) & 0x7fffff
    fl = np.r_[0:21, 21:25, 26:30]       # float words: bodies, sleep, joint impulses
    sub = int(((expo[fl] == 0) & (man[fl] != 0)).sum())
    cnt = [int(u[31 + 16 * k]) for k in range(6)]
    print(f"env {e}: hull ({f[0]:.3f},{f[1]:.3f},a={f[2]:.3f}) v=({f[3]:.3g},{f[4]:.3g},w={f[5]:.3g}) legs a=({f[8]:.3f},{f[14]:.3f}) "
          f"joint states ({u[25]},{u[30]}) imp=({f[21]:.3g},{f[22]:.3g},{f[23]:.3g},{f[24]:.3g}) contacts {cnt} subnormal words {sub} "
          f"nonfinite {int((~np.isfinite(f[fl])).sum())} ep_len {ep_len[e]}")
