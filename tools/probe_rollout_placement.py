#!/usr/bin/env python3
"""Why is the persistent LunarLander rollout's slowest workgroup 45 % above the mean one: placement or env state?

Probe build (`make -C gymrl_amd/csrc prof`, loaded through GYMRL_HIP_LIB).  One whole-rollout launch (4096 envs, T steps);
per workgroup: busy time, the section timers, wave 0's wait at the step barrier, the refill / critic waves' busy time,
HW_ID + XCC_ID of its four waves; per env: the steps on which the env ITSELF had a contact, its own position iterations,
the steps on which its wave ran the contact sweeps.  From these: (a) do two waves of a workgroup share a SIMD, do two
workgroups share a CU, and does either predict the busy time; (b) the cost a step has for the wave (fit over the 256
workgroups) applied to every env's OWN counters = what the slowest env would cost alone — the floor any re-dealing of envs
to workgroups could reach.

usage: probe_rollout_placement.py [T=2048] [train_iters=0]   (train_iters: PPO iterations run first, so that the policy lands)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GYMRL_HIP_LIB", os.path.join(ROOT, "gymrl_amd", "libgymrl_hip_prof.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gymrl_amd.ppo_lunarlander import Config, PPOTrainer  # noqa: E402

N = 4096
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
S = 24
cfg = Config()
cfg.num_envs, cfg.update_freq, cfg.seed, cfg.rollout_chunk = N, T, 0, T
cfg.num_minibatches = 32
sys.stdout = open(os.devnull, "w")
tr = PPOTrainer(cfg)
for _ in range(ITERS):
    tr.update(tr.collect_rollout())
sys.stdout = sys.__stdout__
tr.collect_rollout()
G = N // 16
tr._wg_ticks = torch.zeros(2 * G + S * G + 3 * N, dtype=torch.int64, device=tr.device)
tr.collect_rollout()
torch.cuda.synchronize()
tk = tr._wg_ticks.cpu().numpy()
busy = (tk[1:2 * G:2] - tk[0:2 * G:2]) / 100.0 / T
raw = tk[2 * G:(2 + S) * G].reshape(G, S)
sec = raw / 100.0 / T
env = tk[(2 + S) * G:].reshape(3, N)
own_contact, own_pos, wave_contact = env[0] / T, env[1] / T, env[2] / T
span = (tk[1:2 * G:2].max() - tk[0:2 * G:2].min()) / 100.0 / T
print(f"T={T}, {ITERS} PPO iterations before the probed rollout")
print(f"per-WG busy us/step: mean {busy.mean():.1f} p50 {np.percentile(busy, 50):.1f} p90 {np.percentile(busy, 90):.1f} "
      f"p99 {np.percentile(busy, 99):.1f} max {busy.max():.1f}; launch span {span:.1f}")

# ---------------------------------------------------------------- (a) placement
hw = raw[:, 20:24].astype(np.uint64)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 15
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
cukey = ((xcc[:, 0] * 8 + se[:, 0]) * 2 + sh[:, 0]) * 16 + cu[:, 0]
same_cu = np.array([(xcc[g] == xcc[g, 0]).all() and (se[g] == se[g, 0]).all() and (cu[g] == cu[g, 0]).all() for g in range(G)])
simd_shared0 = np.array([(simd[g, 1:] == simd[g, 0]).any() for g in range(G)])
distinct_simds = np.array([len(set(simd[g].tolist())) for g in range(G)])
uniq, cnt = np.unique(cukey, return_counts=True)
wg_per_cu = dict(zip(uniq.tolist(), cnt.tolist()))
shared_cu = np.array([wg_per_cu[k] > 1 for k in cukey.tolist()])
print("\n(a) placement")
print(f"workgroups whose four waves sit on one CU: {int(same_cu.sum())} / {G};  distinct SIMDs per workgroup: "
      f"{dict(zip(*[x.tolist() for x in np.unique(distinct_simds, return_counts=True)]))}")
print(f"wave 0 shares its SIMD with another wave of its workgroup: {int(simd_shared0.sum())} workgroups; "
      f"CUs in use: {len(uniq)} (workgroups on a CU that holds more than one: {int(shared_cu.sum())})")
print(f"workgroups per XCC: {np.bincount(xcc[:, 0], minlength=8).tolist()}")
for name, m in (("wave 0 shares a SIMD", simd_shared0), ("CU shared with another workgroup", shared_cu)):
    if m.any() and (~m).any():
        print(f"busy us/step where {name}: {busy[m].mean():.1f} (n={int(m.sum())}), elsewhere {busy[~m].mean():.1f}")
    else:
        print(f"busy us/step where {name}: no such workgroup" if not m.any() else f"{name}: every workgroup")
print("busy us/step by XCC: " + " ".join(f"{busy[xcc[:, 0] == x].mean():.1f}" for x in range(8) if (xcc[:, 0] == x).any()))

# ---------------------------------------------------------------- (b) env state
wcs = wave_contact.reshape(G, 16)[:, 0]                # fraction of steps on which the wave ran the contact sweeps
wpos = raw[:, 11] / T                                   # position iterations per step (wave max)
A = np.stack([np.ones(G), wcs, wpos], 1)
coef, *_ = np.linalg.lstsq(A, busy, rcond=None)
res = busy - A @ coef
print("\n(b) env state")
print(f"fit over the {G} workgroups: busy = {coef[0]:.1f} + {coef[1]:.1f} * (share of steps with a contact in the wave) + "
      f"{coef[2]:.2f} * (position iterations per step); residual rms {np.sqrt((res ** 2).mean()):.2f} us, max |res| {np.abs(res).max():.2f}")
own = coef[0] + coef[1] * own_contact + coef[2] * own_pos
tile_of_own = own.reshape(G, 16).max(1)
print(f"every env ALONE (its own contact steps and position iterations through the same fit): mean {own.mean():.1f} "
      f"p99 {np.percentile(own, 99):.1f} max {own.max():.1f} us/step  <- the floor of any re-dealing of envs to workgroups")
print(f"share of steps with a contact: mean env {own_contact.mean():.3f}, mean wave {wcs.mean():.3f}; "
      f"position iterations per step: mean env {own_pos.mean():.2f}, mean wave {wpos.mean():.2f}")
print(f"barrier wait of wave 0 per step: mean {sec[:, 16].mean():.2f} max {sec[:, 16].max():.2f} us; critic wave busy {sec[:, 17].mean():.1f}; "
      f"refill wave busy {sec[:, 18].mean():.1f} us/step, builds a world on {raw[:, 19].sum() / (G * T) * 100:.1f} % of the steps")
order = np.argsort(-busy)
print("\nwg   busy  fit   sweeps (contact-path / contact-free-path)  positions pos-it/step  forward  wait  refill | wave-contact-share  "
      "worst env alone (own contact share, own pos-it) | xcc se cu simd[w0..w3]")
for g in list(order[:12]) + list(order[G // 2:G // 2 + 3]):
    e = 16 * g + int(np.argmax(own[16 * g:16 * g + 16]))
    print(f"{g:4d} {busy[g]:6.1f} {(A @ coef)[g]:6.1f} {sec[g, 2]:7.1f} ({sec[g, 6]:6.1f} / {sec[g, 7]:6.1f}) {sec[g, 4]:9.1f} {wpos[g]:9.2f} "
          f"{sec[g, 8]:8.1f} {sec[g, 16]:5.1f} {sec[g, 18]:6.1f} | {wcs[g]:6.3f}  {own[e]:6.1f} ({own_contact[e]:.3f}, {own_pos[e]:5.2f}) | "
          f"{xcc[g, 0]} {se[g, 0]} {cu[g, 0]:2d} {simd[g].tolist()}")
