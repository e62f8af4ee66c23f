"""TD3 at its default sizes (hidden 256, batch 128): (1) N consecutive TD3Trainer.update() calls against plain
torch autograd + torch.optim.Adam on the same weights, index draws and smoothing noise; (2) what a short train()
leaves in the replay ring (consecutive rows chain, action / reward / done statistics, exploration noise).
GPU box only:  python tools/diag_td3.py [updates]"""
import copy
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gymrl_amd.td3_pendulum import Config, TD3Trainer  # noqa: E402


class PlainActor(nn.Module):
    def __init__(self, src, bound):
        super().__init__()
        self.l = nn.ModuleList([nn.Linear(m.in_features, m.out_features) for m in (src.fc1, src.fc2, src.fc3)])
        for dst, m in zip(self.l, (src.fc1, src.fc2, src.fc3)):
            dst.weight.data.copy_(m.weight.data); dst.bias.data.copy_(m.bias.data)
        self.bound = bound

    def forward(self, x):
        return torch.tanh(self.l[2](F.relu(self.l[1](F.relu(self.l[0](x)))))) * self.bound


class PlainCritic(nn.Module):
    def __init__(self, src):
        super().__init__()
        ms = (src.fc1, src.fc2, src.fc3, src.fc4, src.fc5, src.fc6)
        self.l = nn.ModuleList([nn.Linear(m.in_features, m.out_features) for m in ms])
        for dst, m in zip(self.l, ms):
            dst.weight.data.copy_(m.weight.data); dst.bias.data.copy_(m.bias.data)

    def forward(self, s, a):
        x = torch.cat([s, a], 1)
        return (self.l[2](F.relu(self.l[1](F.relu(self.l[0](x))))), self.l[5](F.relu(self.l[4](F.relu(self.l[3](x))))))

    def q1(self, s, a):
        x = torch.cat([s, a], 1)
        return self.l[2](F.relu(self.l[1](F.relu(self.l[0](x)))))


def maxdiff(plain, ours):
    mods = (ours.fc1, ours.fc2, ours.fc3) + ((ours.fc4, ours.fc5, ours.fc6) if hasattr(ours, "fc4") else ())
    return max(max(float((p.weight - m.weight).abs().max()), float((p.bias - m.bias).abs().max())) for p, m in zip(plain.l, mods))


def check_updates(n_updates):
    cfg = Config(); cfg.seed = 0
    tr = TD3Trainer(cfg)
    dev = tr.device
    g = torch.Generator(device="cpu").manual_seed(1)
    M = 4096
    s = torch.randn(M, 3, generator=g).to(dev); s2 = torch.randn(M, 3, generator=g).to(dev)
    a = (torch.rand(M, 1, generator=g) * 4 - 2).to(dev); r = (-8 * torch.rand(M, generator=g)).to(dev)
    d = (torch.rand(M, generator=g) < 0.05).to(torch.uint8).to(dev)
    tr.memory.push(s, a, r, s2, d)
    pa, pc = PlainActor(tr.actor, tr.action_bound).to(dev), PlainCritic(tr.critic).to(dev)
    pat, pct = copy.deepcopy(pa), copy.deepcopy(pc)
    oa, oc = torch.optim.Adam(pa.parameters(), lr=cfg.lr_actor), torch.optim.Adam(pc.parameters(), lr=cfg.lr_critic)
    for k in range(1, n_updates + 1):
        idx = torch.randint(0, M, (cfg.batch_size,), generator=g).to(torch.int32).to(dev)
        eps = torch.randn(cfg.batch_size, 1, generator=g, dtype=torch.float64).to(dev)
        al, cl = tr.update(indices=idx, eps=eps)
        li = idx.long()
        S, A, R, S2, D = s[li], a[li], r[li].unsqueeze(1), s2[li], d[li].float().unsqueeze(1)
        with torch.no_grad():
            nz = (eps.float() * cfg.policy_noise).clamp(-cfg.noise_clip, cfg.noise_clip)
            na = (pat(S2) + nz).clamp(-tr.action_bound, tr.action_bound)
            t1, t2 = pct(S2, na)
            y = R + cfg.gamma * (1 - D) * torch.min(t1, t2)
        q1, q2 = pc(S, A)
        closs = F.mse_loss(q1, y) + F.mse_loss(q2, y)
        oc.zero_grad(); closs.backward(); oc.step()
        aloss = 0.0
        if k % cfg.policy_freq == 0:
            al_t = -pc.q1(S, pa(S)).mean()
            oa.zero_grad(); al_t.backward(); oa.step()
            with torch.no_grad():
                for tgt, src in ((pat, pa), (pct, pc)):
                    for tp, sp in zip(tgt.parameters(), src.parameters()):
                        tp.copy_(cfg.tau * sp + (1 - cfg.tau) * tp)
            aloss = float(al_t)
        if k <= 4 or k % 50 == 0 or k == n_updates:
            print(f"update {k}: losses ours ({al:.6f}, {cl:.6f}) torch ({aloss:.6f}, {float(closs):.6f})  max|dW| actor "
                  f"{maxdiff(pa, tr.actor):.2e} critic {maxdiff(pc, tr.critic):.2e} actor_target {maxdiff(pat, tr.actor_target):.2e} "
                  f"critic_target {maxdiff(pct, tr.critic_target):.2e}", flush=True)


def check_loop(steps):
    cfg = Config(); cfg.seed = 0; cfg.max_episodes = 10 ** 9
    tr = TD3Trainer(cfg)
    tr.train(max_vector_steps=steps)
    n = len(tr.memory)
    s, a, r, s2, d = (t[:n] for t in tr.memory.ring)
    a = a.view(torch.float32)
    chain = (s2[:-1] - s[1:]).abs().max(dim=1).values
    notdone = d[:-1] == 0
    print(f"loop: rows {n}; done rows {int(d.sum())} at {torch.nonzero(d).flatten()[:5].tolist()}; "
          f"next_state[i] == state[i+1] on non-done rows: max diff {float(chain[notdone].max()):.3g}; "
          f"on done rows mean diff {float(chain[~notdone].mean()) if (~notdone).any() else 0:.3g}")
    print(f"loop: action mean {float(a.mean()):.3f} std {float(a.std()):.3f} min {float(a.min()):.3f} max {float(a.max()):.3f}; "
          f"|a| == 2 on {float((a.abs() >= 2).float().mean()):.3f} of rows; reward mean {float(r.mean()):.3f} min {float(r.min()):.3f}")
    th = torch.atan2(s[:, 1], s[:, 0])
    cost = th ** 2 + 0.1 * s[:, 2] ** 2 + 0.001 * a[:, 0] ** 2
    print(f"loop: reward + cost(state, action) max abs {float((r + cost).abs().max()):.3g}; |obs| norm max dev "
          f"{float(((s[:, 0] ** 2 + s[:, 1] ** 2) - 1).abs().max()):.3g}; thdot range [{float(s[:, 2].min()):.2f}, {float(s[:, 2].max()):.2f}]")
    with torch.no_grad():
        mu = tr.actor(s)
    nz = a - mu
    print(f"loop: (stored action - current actor(s)) std {float(nz.std()):.3f}; episode rewards {[round(x) for x in list(tr.episode_rewards)[-8:]]}")
    # exploration noise of consecutive select_action calls on one fixed state
    st = s[:1].clone()
    det = float(tr.select_action(st, deterministic=True))
    draws = torch.tensor([float(tr.select_action(st)) - det for _ in range(2000)])
    print(f"noise: 2000 draws mean {float(draws.mean()):.4f} std {float(draws.std()):.4f} (want {cfg.exploration_noise * tr.action_bound:.3f}) "
          f"lag-1 corr {float(torch.corrcoef(torch.stack([draws[:-1], draws[1:]]))[0, 1]):.3f}")
    idx = torch.cat([tr.memory.draw_indices(128).long().cpu() for _ in range(200)])
    print(f"draws: 25600 indices over {n} rows: min {int(idx.min())} max {int(idx.max())} mean/size {float(idx.float().mean()) / n:.3f} "
          f"distinct {idx.unique().numel()}")


if __name__ == "__main__":
    check_updates(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    check_loop(3000)
