#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python learner code.

Runs only in the build container (needs /root/reference).  It imports the
reference scripts behind a stub `gymnasium` (the scripts touch gym only through
gym.make), drives their functions on seeded inputs, and stores inputs + expected
outputs as small .npz fixtures next to this file.  No reference source is copied:
the fixtures are data.  The fixtures pin oracle/gymrl_oracle.c
(tests/test_oracle_golden.py), which in turn checks the HIP kernels on the GPU.

    python tests/golden/make_golden.py            # regenerate everything
"""
import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------- stub gym ----
class _Space:
    def __init__(self, shape=None, n=None, high=None):
        self.shape, self.n = shape, n
        self.high = None if high is None else np.array(high, np.float32)

    def sample(self):
        return random.randrange(self.n)


class _FakeEnv:
    """Shape-only env: the golden generators never step it."""

    def __init__(self, name):
        if name.startswith("LunarLander"):
            self.observation_space, self.action_space, steps = _Space((8,)), _Space(n=4), 1000
        elif name.startswith("CartPole"):
            self.observation_space, self.action_space, steps = _Space((4,)), _Space(n=2), 500
        else:
            self.observation_space, self.action_space, steps = _Space((3,)), _Space((1,), high=[2.0]), 200
        self.spec = types.SimpleNamespace(max_episode_steps=steps)

    def reset(self, seed=None):
        return np.zeros(self.observation_space.shape, np.float32), {}

    def close(self):
        pass


def _install_stub_gym():
    g = types.ModuleType("gymnasium")
    g.make = lambda name, **kw: _FakeEnv(name)
    sys.modules["gymnasium"] = g


def load_ref(relpath, modname):
    _install_stub_gym()
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)} B)")


# ------------------------------------------------------------------- G1 ------
def gen_gae():
    """PPOTrainer.compute_gae (ppo_lunarlander.py:179-196) on single-env rollouts."""
    ppo = load_ref("algorithms/ppo_lunarlander.py", "ref_ppo")
    rng = np.random.default_rng(1)
    cases = {}
    k = 0
    for T in (1, 7, 64, 2048):
        for pattern in ("none", "all", "last", "random"):
            rew = rng.normal(size=T) * np.where(rng.random(T) < 0.01, 100.0, 1.0)
            val = rng.normal(size=T).astype(np.float32)
            if pattern == "none":
                done = np.zeros(T, bool)
            elif pattern == "all":
                done = np.ones(T, bool)
            elif pattern == "last":
                done = np.zeros(T, bool)
                done[-1] = True
            else:
                done = rng.random(T) < 0.05
            next_value = float(np.float32(rng.normal()))
            tr = ppo.PPOTrainer.__new__(ppo.PPOTrainer)
            tr.cfg = ppo.Config()
            tr.buffer = ppo.RolloutBuffer()
            # rewards in the reference are python floats that came from float32 env values
            rew32 = rew.astype(np.float32)
            tr.buffer.rewards = [float(x) for x in rew32]
            tr.buffer.values = [float(x) for x in val]
            tr.buffer.dones = [bool(x) for x in done]
            adv, ret = tr.compute_gae(next_value)
            assert adv.dtype == np.float64
            norm = (adv - adv.mean()) / (adv.std() + 1e-8)      # :236
            cases[f"c{k}_rew"] = rew32
            cases[f"c{k}_val"] = val
            cases[f"c{k}_done"] = done.astype(np.uint8)
            cases[f"c{k}_next"] = np.float32(next_value)
            cases[f"c{k}_adv"] = adv
            cases[f"c{k}_ret"] = ret
            cases[f"c{k}_norm"] = norm
            k += 1
    cases["n_cases"] = np.int64(k)
    cases["gamma"] = np.float64(ppo.Config().gamma)
    cases["lam"] = np.float64(ppo.Config().gae_lambda)
    save("gae_g1", **cases)


def gen_gae_g2():
    """ReplayBuffer_on_policy.compute_advantage (utils/buffer.py:21-35)."""
    sys.path.insert(0, REF)
    from utils.buffer import ReplayBuffer_on_policy
    rng = np.random.default_rng(2)
    cfg = types.SimpleNamespace(gamma=0.99, lamda=0.95, device="cpu")
    cases = {}
    k = 0
    for T in (2, 9, 200):
        for _ in range(2):
            rew = rng.normal(size=(T, 1)).astype(np.float32)
            val = rng.normal(size=(T, 1)).astype(np.float32)
            nval = rng.normal(size=(T, 1)).astype(np.float32)
            done = (rng.random((T, 1)) < 0.1)
            dw = done & (rng.random((T, 1)) < 0.5)
            buf = ReplayBuffer_on_policy(cfg)
            adv, vt = buf.compute_advantage(torch.tensor(rew), torch.tensor(done.astype(np.float32)),
                                            torch.tensor(dw.astype(np.float32)), torch.tensor(val),
                                            torch.tensor(nval))
            for nm, a in (("rew", rew), ("val", val), ("nval", nval), ("done", done.astype(np.uint8)),
                          ("dw", dw.astype(np.uint8)), ("advn", adv.numpy()), ("vt", vt.numpy())):
                cases[f"c{k}_{nm}"] = a
            k += 1
    cases["n_cases"] = np.int64(k)
    cases["gamma"] = np.float64(cfg.gamma)
    cases["lam"] = np.float64(cfg.lamda)
    save("gae_g2", **cases)


def gen_gae_g3():
    """ppo_full compute_advantages (ppo_full_lunarlander.py:507-535), lam_actor != lam_critic."""
    pf = load_ref("algorithms/ppo_full_lunarlander.py", "ref_ppo_full")
    rng = np.random.default_rng(3)
    cases = {}
    k = 0
    for T in (5, 333):
        rew = rng.normal(size=T).astype(np.float32)
        val = rng.normal(size=T).astype(np.float32)
        done = rng.random(T) < 0.03
        nv = float(np.float32(rng.normal()))
        tr = pf.PPOTrainer.__new__(pf.PPOTrainer)
        tr.cfg = pf.Config()
        tr.cfg.lam_actor, tr.cfg.lam_critic = 0.9, 0.97
        tr.buffer = pf.RolloutBuffer()
        tr.buffer.rewards = [float(x) for x in rew]
        tr.buffer.values = [float(x) for x in val]
        tr.buffer.dones = [bool(x) for x in done]
        tr.buffer.next_value = nv
        adv, ret = tr.compute_advantages()
        for nm, a in (("rew", rew), ("val", val), ("done", done.astype(np.uint8)), ("next", np.float32(nv)),
                      ("adv", np.asarray(adv, np.float64)), ("ret", np.asarray(ret, np.float64))):
            cases[f"c{k}_{nm}"] = a
        k += 1
    cases["n_cases"] = np.int64(k)
    cases["gamma"] = np.float64(tr.cfg.gamma)
    cases["lam_actor"], cases["lam_critic"] = np.float64(0.9), np.float64(0.97)
    save("gae_g3", **cases)


# ------------------------------------------------------------------- P2 ------
def gen_categorical():
    """ActorCritic.get_action's Categorical (ppo_lunarlander.py:92-104): action under
    a torch seed + the Exp(1) draw torch.multinomial consumed under the same seed."""
    rng = np.random.default_rng(4)
    from torch.distributions import Categorical
    n, A = 512, 4
    logits = (rng.normal(size=(n, A)) * rng.choice([0.1, 1.0, 5.0], size=(n, 1))).astype(np.float32)
    lt = torch.tensor(logits)
    torch.manual_seed(1234)
    dist = Categorical(logits=lt)
    action = dist.sample()
    logp = dist.log_prob(action)
    ent = dist.entropy()
    torch.manual_seed(1234)
    q = torch.empty_like(dist.probs).exponential_(1.0)
    assert torch.equal((dist.probs / q).argmax(-1), action), "multinomial != argmax(p/q)"
    save("categorical", logits=logits, noise_exp=q.numpy(), action=action.numpy().astype(np.int32),
         logp=logp.numpy(), entropy=ent.numpy(), argmax=lt.argmax(-1).numpy().astype(np.int32))


# ------------------------------------------------------------- L1 / O1 -------
def gen_ppo_loss():
    """One reference minibatch: loss terms, metrics, dlogits/dvalue (autograd hooks),
    parameter grads, clipped-norm Adam step (ppo_lunarlander.py:274-322)."""
    ppo = load_ref("algorithms/ppo_lunarlander.py", "ref_ppo")
    cfg = ppo.Config()
    out = {}
    for case, B in enumerate((64, 257)):
        seed_all(10 + case)
        model = ppo.ActorCritic(8, 4, 32)  # small hidden: fixture stays a few hundred KB
        opt = torch.optim.Adam(model.parameters(), lr=cfg.lr, eps=1e-5)
        states = torch.randn(B, 8)
        with torch.no_grad():
            lg, _ = model(states)
            # behaviour policy = perturbed current policy so ~10% of ratios clip
            old_logits = lg + 0.6 * torch.randn_like(lg)
            d_old = torch.distributions.Categorical(logits=old_logits)
            actions = d_old.sample()
            old_lp = d_old.log_prob(actions)
        adv = torch.randn(B) * 2.0
        ret = torch.randn(B) * 3.0
        params0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().copy()

        logits, values = model(states)
        logits.retain_grad()
        values.retain_grad()
        dist = torch.distributions.Categorical(logits=logits)
        new_lp = dist.log_prob(actions)
        entropy = dist.entropy()
        v = values.squeeze(-1)
        ratio = torch.exp(new_lp - old_lp)
        surr1 = ratio * adv
        surr2 = torch.clamp(ratio, 1 - cfg.clip_eps, 1 + cfg.clip_eps) * adv
        min_surr = torch.min(surr1, surr2)
        policy_loss = -torch.mean(torch.where(adv < 0, torch.max(min_surr, cfg.dual_clip * adv), min_surr))
        value_loss = cfg.value_coef * torch.mean((v - ret).pow(2))
        entropy_loss = -cfg.entropy_coef * entropy.mean()
        loss = policy_loss + value_loss + entropy_loss
        opt.zero_grad()
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy().copy()
        total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.max_grad_norm)
        opt.step()
        params1 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().copy()
        st = [opt.state[p] for p in model.parameters()]
        m1 = torch.cat([s["exp_avg"].reshape(-1) for s in st]).numpy().copy()
        v1 = torch.cat([s["exp_avg_sq"].reshape(-1) for s in st]).numpy().copy()
        clip_frac = ((ratio < 1 - cfg.clip_eps) | (ratio > 1 + cfg.clip_eps)).float().mean().item()
        kl = (old_lp - new_lp).mean().item()
        pre = f"c{case}_"
        out.update({
            pre + "logits": logits.detach().numpy(), pre + "values": v.detach().numpy(),
            pre + "actions": actions.numpy().astype(np.int32), pre + "old_lp": old_lp.numpy(),
            pre + "adv": adv.numpy(), pre + "ret": ret.numpy(),
            pre + "dlogits": logits.grad.numpy(), pre + "dvalues": values.grad.squeeze(-1).numpy(),
            pre + "metrics": np.array([policy_loss.item(), value_loss.item(), entropy.mean().item(),
                                       clip_frac, kl], np.float64),
            pre + "params0": params0, pre + "grads": grads, pre + "total_norm": np.float64(total_norm.item()),
            pre + "params1": params1, pre + "m1": m1, pre + "v1": v1,
        })
    out["n_cases"] = np.int64(2)
    out["cfg"] = np.array([cfg.clip_eps, cfg.dual_clip, cfg.value_coef, cfg.entropy_coef], np.float64)
    out["adam"] = np.array([cfg.lr, 0.9, 0.999, 1e-5, cfg.max_grad_norm], np.float64)
    save("ppo_loss", **out)


def gen_adam_multi():
    """torch.optim.Adam over several steps with and without clipping / grad clamp
    (ppo_lunarlander.py:302-307, dqn_cartpole.py:163-166)."""
    seed_all(20)
    n = 1003
    out = {}
    for case, (lr, eps, max_norm, clamp) in enumerate([(3e-4, 1e-5, 0.5, 0.0), (1e-3, 1e-8, 0.0, 1.0),
                                                       (1e-3, 1e-8, 10.0, 0.0)]):
        p = torch.nn.Parameter(torch.randn(n))
        opt = torch.optim.Adam([p], lr=lr, eps=eps)
        p0 = p.detach().numpy().copy()
        gs = []
        for step in range(5):
            g = torch.randn(n) * (3.0 if step % 2 == 0 else 0.01)
            gs.append(g.numpy().copy())
            p.grad = g.clone()
            if clamp > 0:
                p.grad.data.clamp_(-clamp, clamp)
            if max_norm > 0:
                torch.nn.utils.clip_grad_norm_([p], max_norm)
            opt.step()
        out[f"c{case}_p0"] = p0
        out[f"c{case}_grads"] = np.stack(gs)
        out[f"c{case}_p5"] = p.detach().numpy().copy()
        out[f"c{case}_m5"] = opt.state[p]["exp_avg"].numpy().copy()
        out[f"c{case}_v5"] = opt.state[p]["exp_avg_sq"].numpy().copy()
        out[f"c{case}_hp"] = np.array([lr, 0.9, 0.999, eps, max_norm, clamp], np.float64)
    out["n_cases"] = np.int64(3)
    save("adam", **out)


# ------------------------------------------------------------------- L3 ------
def gen_ppo_full_loss():
    """ppo_full update_model minibatch loss (ppo_full_lunarlander.py:575-652) on given
    logits/values (the mHC network itself runs through PyTorch and is not restated)."""
    pf = load_ref("algorithms/ppo_full_lunarlander.py", "ref_ppo_full")
    cfg = pf.Config()
    from torch.distributions import Categorical
    out = {}
    for case, B in enumerate((128, 1024, 512)):
        seed_all(30 + case)
        cov_ratio, cov_min, cov_max = (0.5, 0.2, 5.0) if case == 2 else (cfg.clip_cov_ratio, cfg.clip_cov_min, cfg.clip_cov_max)
        logits = (torch.randn(B, 4) * 1.5).requires_grad_(True)
        values = torch.randn(B, 1, requires_grad=True)
        with torch.no_grad():
            old_logits = logits + 0.4 * torch.randn(B, 4)
            d_old = Categorical(logits=old_logits)
            a_batch = d_old.sample()
            old_lp = d_old.log_prob(a_batch)
            old_ent = d_old.entropy()
        adv_batch = torch.randn(B) * 2
        ret_batch = torch.randn(B) * 3
        ent_coef = 0.0077
        dist = Categorical(logits=logits)
        new_lp = dist.log_prob(a_batch)
        new_ent = dist.entropy()
        entropy_ratio = new_ent / (old_ent + 1e-8)
        erc_mask = ((entropy_ratio > (1 - cfg.erc_beta_low)) & (entropy_ratio < (1 + cfg.erc_beta_high))).float()
        ratio = (new_lp - old_lp).exp()
        covs = (new_lp - new_lp.mean()) * (adv_batch - adv_batch.mean())
        corr = torch.ones_like(adv_batch) * erc_mask
        clip_ratio = ratio.clamp(0.0, cfg.dual_clip)
        surr1 = clip_ratio * adv_batch
        surr2 = torch.clamp(ratio, 1 - cfg.clip_eps_min, 1 + cfg.clip_eps_max) * adv_batch
        cov_perm = np.zeros(0, np.int64)
        clip_idx = torch.where((covs > cov_min) & (covs < cov_max))[0]                       # :611-616
        if len(clip_idx) > 0 and cov_ratio > 0:
            clip_num = max(int(len(clip_idx) * cov_ratio), 1)
            perm_ = torch.randperm(len(clip_idx))
            cov_perm = perm_.numpy().copy()
            clip_idx = clip_idx[perm_[: min(clip_num, len(clip_idx))]]
            corr[clip_idx] = 0.0
        clip_frac = torch.mean(((ratio < (1 - cfg.clip_eps_min)) | (ratio > (1 + cfg.clip_eps_max))).float() * corr)
        policy_loss = torch.mean(-torch.min(surr1, surr2) * corr)
        value_loss = torch.mean(0.5 * corr * (values.squeeze() - ret_batch).pow(2))
        entropy = (dist.entropy() * corr).mean()
        entropy_loss = torch.mean(-ent_coef * entropy)
        loss = policy_loss + value_loss + entropy_loss
        loss.backward()
        pre = f"c{case}_"
        out.update({
            pre + "logits": logits.detach().numpy(), pre + "values": values.detach().squeeze(-1).numpy(),
            pre + "actions": a_batch.numpy().astype(np.int32), pre + "old_lp": old_lp.numpy(),
            pre + "old_ent": old_ent.numpy(), pre + "adv": adv_batch.numpy(), pre + "ret": ret_batch.numpy(),
            pre + "dlogits": logits.grad.numpy(), pre + "dvalues": values.grad.squeeze(-1).numpy(),
            pre + "cov_perm": cov_perm, pre + "cov_cfg": np.array([cov_ratio, cov_min, cov_max], np.float64),
            pre + "corr": corr.numpy(),
            pre + "metrics": np.array([policy_loss.item(), value_loss.item(), entropy.item(), clip_frac.item(),
                                       (old_lp - new_lp).mean().item(), 1.0 - erc_mask.mean().item(),
                                       covs.mean().item()], np.float64),
        })
    out["n_cases"] = np.int64(3)
    out["cfg"] = np.array([cfg.clip_eps_min, cfg.clip_eps_max, cfg.dual_clip, cfg.erc_beta_low,
                           cfg.erc_beta_high, 0.0077], np.float64)
    save("ppo_full_loss", **out)


def gen_soft_update():
    """SACTrainer.soft_update arithmetic (sac_pendulum.py:194-199) on flat tensors."""
    seed_all(40)
    tgt, src = torch.randn(777), torch.randn(777)
    tau = 0.005
    new = tau * src + (1.0 - tau) * tgt
    save("soft_update", target=tgt.numpy(), source=src.numpy(), tau=np.float64(tau), out=new.numpy())


# ============================================================== off-policy ====
def gen_sumtree():
    """SumTree (rainbow_dqn_cartpole.py:116-152): scripted update/get_index/priority_max traces,
    capacity 20 (the reference default's shape: not a power of two) and 16."""
    rb = load_ref("algorithms/rainbow_dqn_cartpole.py", "ref_rainbow")
    rng = np.random.default_rng(50)
    out = {}
    for case, cap in enumerate((20, 16, 5)):
        tree = rb.SumTree(cap)
        ops_idx, ops_p, snaps = [], [], []
        for rnd in range(6):
            B = int(rng.integers(1, 2 * cap))
            idx = rng.integers(0, cap, size=B)
            pr = rng.random(B) * 3 + 0.01
            for i, p in zip(idx, pr):
                tree.update(int(i), float(p))
            ops_idx.append(np.pad(idx, (0, 2 * cap - B), constant_values=-1))
            ops_p.append(np.pad(pr, (0, 2 * cap - B)))
            snaps.append(tree.tree.copy())
        vs = rng.random(64) * tree.priority_sum
        gi = np.array([tree.get_index(float(v)) for v in vs])
        out[f"c{case}_cap"] = np.int64(cap)
        out[f"c{case}_ops_idx"] = np.stack(ops_idx).astype(np.int32)
        out[f"c{case}_ops_p"] = np.stack(ops_p)
        out[f"c{case}_snaps"] = np.stack(snaps)
        out[f"c{case}_v"] = vs
        out[f"c{case}_get_idx"] = gi[:, 0].astype(np.int64)
        out[f"c{case}_get_prio"] = gi[:, 1]
        out[f"c{case}_max"] = np.float64(tree.priority_max)
    out["n_cases"] = np.int64(3)
    save("sumtree", **out)


def gen_per_nstep():
    """PrioritizedNStepBuffer (rainbow_dqn_cartpole.py:155-264): store_transition stream with
    dones at every window position, then sample (with the uniforms numpy consumed) and
    update_priorities; full buffer + tree snapshots."""
    rb = load_ref("algorithms/rainbow_dqn_cartpole.py", "ref_rainbow")
    cfg = rb.Config()
    cfg.memory_capacity, cfg.batch_size, cfg.device = 48, 16, "cpu"
    buf = rb.PrioritizedNStepBuffer(cfg, 4)
    rng = np.random.default_rng(51)
    T = 70
    obs = rng.normal(size=(T + 1, 4)).astype(np.float32)
    act = rng.integers(0, 2, size=T)
    rew = rng.normal(size=T).astype(np.float32)
    done = rng.random(T) < 0.2
    term = done & (rng.random(T) < 0.7)
    for t in range(T):
        buf.store_transition(obs[t], int(act[t]), float(rew[t]), obs[t + 1], bool(term[t]), bool(done[t]))
    tree_after_store = buf.sum_tree.tree.copy()
    state_arr = {k: v.copy() for k, v in buf.buffer.items()}
    np.random.seed(777)
    batch, index, w = buf.sample(300, 1000)
    np.random.seed(777)
    u = np.random.random_sample(cfg.batch_size)
    td = rng.normal(size=cfg.batch_size).astype(np.float32)   # td_error.detach().cpu().numpy() is float32 (:340)
    td[3] = td[7]                      # exercise equal priorities
    index2 = index.copy()
    index2[5] = index2[2]              # and a duplicate leaf inside one batch
    buf.update_priorities(index2, td)
    save("per_nstep", obs=obs, act=act.astype(np.int32), rew=rew, done=done.astype(np.uint8),
         term=term.astype(np.uint8), cap=np.int64(cfg.memory_capacity), n_steps=np.int64(cfg.n_steps),
         gamma=np.float64(cfg.gamma), alpha=np.float64(cfg.alpha), beta=np.float64(buf.beta),
         size=np.int64(buf.current_size), count=np.int64(buf.count),
         buf_state=state_arr["state"], buf_action=state_arr["action"], buf_reward=state_arr["reward"],
         buf_next=state_arr["next_state"], buf_terminal=state_arr["terminal"], tree_after_store=tree_after_store,
         u=u, index=index.astype(np.int64), is_weight=w.numpy(), batch_reward=batch["reward"].numpy(),
         batch_state=batch["state"].numpy(), td=td, index2=index2.astype(np.int64),
         tree_after_update=buf.sum_tree.tree.copy())


def gen_per_variant_b():
    """ddqn_per_cartpole.py:67-147 (PER variant B): push/sample/update_priorities trace."""
    pb = load_ref("algorithms/ddqn_per_cartpole.py", "ref_ddqn_per")
    cfg = pb.Config()
    cfg.memory_capacity, cfg.batch_size = 32, 8
    beta0 = cfg.beta
    buf = pb.PrioritizedReplayBuffer(cfg)
    rng = np.random.default_rng(52)
    for t in range(45):
        buf.push((np.float32(t), 0, 0.0, np.float32(t + 1), False))
    tree_after_push = buf.tree.tree.copy()
    random.seed(99)
    _, indices, w = buf.sample(cfg.batch_size)
    random.seed(99)
    u = np.array([random.random() for _ in range(cfg.batch_size)])
    errs = (np.abs(rng.normal(size=cfg.batch_size)) * 2).astype(np.float32)   # td_errors.abs()...numpy() is float32
    buf.update_priorities(indices, errs)
    save("per_variant_b", cap=np.int64(32), n_push=np.int64(45), tree_after_push=tree_after_push, u=u,
         indices=indices.astype(np.int64), is_weight=np.asarray(w, np.float64), errs=errs,
         tree_after_update=buf.tree.tree.copy(), beta=np.float64(cfg.beta), beta0=np.float64(beta0),
         size=np.int64(buf.tree.size), alpha=np.float64(cfg.alpha), eps=np.float64(cfg.eps),
         error_max=np.float64(cfg.error_max))


def gen_noisy():
    """NoisyLinear.reset_noise (rainbow_dqn_cartpole.py:77-87): raw randn draws -> epsilons."""
    rb = load_ref("algorithms/rainbow_dqn_cartpole.py", "ref_rainbow")
    out = {}
    for case, (nin, nout) in enumerate(((256, 2), (256, 1), (7, 5))):
        layer = rb.NoisyLinear(nin, nout)
        torch.manual_seed(60 + case)
        layer.reset_noise()
        torch.manual_seed(60 + case)
        raw_in = torch.randn(nin)
        raw_out = torch.randn(nout)
        out[f"c{case}_raw_in"], out[f"c{case}_raw_out"] = raw_in.numpy(), raw_out.numpy()
        out[f"c{case}_w_eps"], out[f"c{case}_b_eps"] = layer.weight_epsilon.numpy().copy(), layer.bias_epsilon.numpy().copy()
    out["n_cases"] = np.int64(3)
    save("noisy", **out)


def gen_dqn_update():
    """One DQNTrainer.update() (dqn_cartpole.py:135-168) and one RainbowDQNTrainer.update()
    (rainbow_dqn_cartpole.py:311-361) on a memory whose whole content is the batch."""
    dq = load_ref("algorithms/dqn_cartpole.py", "ref_dqn")
    cfg = dq.Config()
    cfg.device, cfg.batch_size, cfg.hidden_dim = "cpu", 32, 32
    seed_all(70)
    tr = dq.DQNTrainer(cfg)
    rng = np.random.default_rng(70)
    trans = []
    for i in range(cfg.batch_size):
        trans.append((rng.normal(size=4).astype(np.float32), int(rng.integers(0, 2)), float(rng.normal()),
                      rng.normal(size=4).astype(np.float32), bool(rng.random() < 0.2)))
        tr.memory.push(*trans[-1])
    with torch.no_grad():   # de-correlate target from policy
        for p in tr.target_net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    sd_policy = {k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()}
    sd_target = {k: v.numpy().copy() for k, v in tr.target_net.state_dict().items()}
    random.seed(5)
    loss = tr.update()
    random.seed(5)
    order = random.sample(range(cfg.batch_size), cfg.batch_size)     # same draw as random.sample(deque, B)
    out = dict(order=np.array(order, np.int32), loss=np.float64(loss),
               states=np.stack([t[0] for t in trans]), actions=np.array([t[1] for t in trans], np.int32),
               rewards=np.array([t[2] for t in trans], np.float32), next_states=np.stack([t[3] for t in trans]),
               dones=np.array([t[4] for t in trans], np.uint8), gamma=np.float64(cfg.gamma), lr=np.float64(cfg.lr))
    for k, v in sd_policy.items():
        out["p0_" + k] = v
    for k, v in sd_target.items():
        out["t0_" + k] = v
    for k, v in tr.policy_net.state_dict().items():
        out["p1_" + k] = v.numpy().copy()
    save("dqn_update", **out)


def gen_sac():
    """Actor.sample forward/backward (sac_pendulum.py:76-87) + one SACTrainer.update() (:213-267)."""
    sac = load_ref("algorithms/sac_pendulum.py", "ref_sac")
    cfg = sac.Config()
    cfg.device, cfg.batch_size, cfg.hidden_dim = "cpu", 24, 32
    seed_all(80)
    tr = sac.SACTrainer(cfg)
    states = torch.randn(64, 3)
    mean, log_std = tr.actor.forward(states)
    mean = (mean * 3).detach().requires_grad_(True)          # spread over tanh's range
    log_std = (log_std.detach() - 1.0 + torch.randn(64, 1)).clamp(cfg.log_std_min, cfg.log_std_max).requires_grad_(True)
    torch.manual_seed(81)
    eps = torch.randn(64, 1)
    std = log_std.exp()
    torch.manual_seed(81)
    normal = torch.distributions.Normal(mean, std)
    x_t = normal.rsample()
    action = torch.tanh(x_t) * tr.action_bound
    log_prob = normal.log_prob(x_t)
    log_prob = log_prob - torch.log(tr.action_bound * (1 - torch.tanh(x_t).pow(2)) + 1e-6)
    log_prob = log_prob.sum(dim=1, keepdim=True)
    assert torch.allclose(x_t, mean + std * eps)
    ga, gl = torch.randn(64, 1), torch.randn(64, 1)
    ((action * ga).sum() + (log_prob * gl).sum()).backward()
    out = dict(mean=mean.detach().numpy(), log_std=log_std.detach().numpy(), eps=eps.numpy(),
               action=action.detach().numpy(), logp=log_prob.detach().numpy()[:, 0], g_action=ga.numpy(),
               g_logp=gl.numpy()[:, 0], d_mean=mean.grad.numpy(), d_log_std=log_std.grad.numpy(),
               bound=np.float64(tr.action_bound))
    # ---- one full update() on a memory that is exactly one batch ----
    rng = np.random.default_rng(82)
    trans = []
    for i in range(cfg.batch_size):
        trans.append((rng.normal(size=3).astype(np.float32), rng.uniform(-2, 2, size=1).astype(np.float32),
                      float(rng.normal()), rng.normal(size=3).astype(np.float32), bool(rng.random() < 0.15)))
        tr.memory.push(*trans[-1])
    with torch.no_grad():
        for p in tr.critic_target.parameters():
            p.add_(0.05 * torch.randn_like(p))
    for name, net in (("actor", tr.actor), ("critic", tr.critic), ("critic_target", tr.critic_target)):
        for k, v in net.state_dict().items():
            out[f"u0_{name}_{k}"] = v.numpy().copy()
    random.seed(9)
    torch.manual_seed(83)
    al, cl, aal = tr.update()
    random.seed(9)
    order = random.sample(range(cfg.batch_size), cfg.batch_size)
    torch.manual_seed(83)
    eps_next = torch.randn(cfg.batch_size, 1)        # actor.sample(next_states)
    eps_cur = torch.randn(cfg.batch_size, 1)         # actor.sample(states)
    for name, net in (("actor", tr.actor), ("critic", tr.critic), ("critic_target", tr.critic_target)):
        for k, v in net.state_dict().items():
            out[f"u1_{name}_{k}"] = v.numpy().copy()
    out.update(u_order=np.array(order, np.int32), u_losses=np.array([al, cl, aal], np.float64),
               u_states=np.stack([t[0] for t in trans]), u_actions=np.stack([t[1] for t in trans]),
               u_rewards=np.array([t[2] for t in trans], np.float32), u_next_states=np.stack([t[3] for t in trans]),
               u_dones=np.array([t[4] for t in trans], np.uint8), u_eps_next=eps_next.numpy(), u_eps_cur=eps_cur.numpy(),
               u_log_alpha0=np.float64(np.log(cfg.init_alpha)), u_log_alpha1=np.float64(tr.log_alpha.item()),
               u_gamma=np.float64(cfg.gamma), u_tau=np.float64(cfg.tau), u_lr=np.float64(cfg.lr_actor),
               u_target_entropy=np.float64(tr.target_entropy))
    save("sac", **out)


def gen_normalization():
    """utils/normalization.py: RunningMeanStd / Normalization / RewardScaling traces (incl. the
    n == 1 special case and a negative first reward)."""
    sys.path.insert(0, REF)
    from utils.normalization import Normalization, RewardScaling
    rng = np.random.default_rng(90)
    x = (rng.normal(size=(40, 8)) * np.array([1, 2, 0.5, 3, 1, 1, 0.1, 10]) + 0.3).astype(np.float32)
    norm = Normalization(shape=8)
    ys = np.stack([np.asarray(norm(x[i]), np.float64) for i in range(40)])
    y_eval = np.asarray(norm(x[3], update=False), np.float64)
    rs = RewardScaling(shape=1, gamma=0.99)
    r = rng.normal(size=60).astype(np.float32)
    r[0] = -3.0
    done = rng.random(60) < 0.1
    outs = []
    for i in range(60):
        outs.append(float(rs(float(r[i]))[0]))
        if done[i]:
            rs.reset()
    save("normalization", x=x, y=ys, y_eval=y_eval, mean=np.asarray(norm.running_ms.mean, np.float64),
         S=np.asarray(norm.running_ms.S, np.float64), std=np.asarray(norm.running_ms.std, np.float64),
         n=np.int64(norm.running_ms.n), r=r, done=done.astype(np.uint8), r_scaled=np.array(outs, np.float64),
         gamma=np.float64(0.99))


def gen_ppo_full_net():
    """ppo_full ActorCritic (mHC backbone, ppo_full_lunarlander.py:364-412) forward + parameter
    gradients at mhc_dim=32 (small fixture); the w / alpha / beta parameters are perturbed so the
    gates and the Sinkhorn mixing are exercised away from their init."""
    pf = load_ref("algorithms/ppo_full_lunarlander.py", "ref_ppo_full")
    cfg = pf.Config()
    cfg.mhc_dim = 32
    seed_all(95)
    net = pf.ActorCritic(8, 4, config=cfg)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith(".w") or n_.endswith(".alpha") or n_.endswith(".beta") or n_.endswith("norm.weight"):
                p_.add_(0.3 * torch.randn_like(p_))
    x = torch.randn(16, 8)
    w1, w2 = torch.randn(16, 4), torch.randn(16, 1)
    logits, values = net(x)
    ((logits * w1).sum() + (values * w2).sum()).backward()
    out = {"x": x.numpy(), "w1": w1.numpy(), "w2": w2.numpy(), "logits": logits.detach().numpy(),
           "values": values.detach().numpy()}
    for k, v in net.state_dict().items():
        out["sd_" + k] = v.numpy().copy()
    for k, p_ in net.named_parameters():
        out["grad_" + k] = p_.grad.numpy().copy()
    save("ppo_full_net", **out)


GENERATORS = [gen_ppo_full_net, gen_gae, gen_gae_g2, gen_gae_g3, gen_categorical, gen_ppo_loss, gen_adam_multi,
              gen_ppo_full_loss, gen_soft_update, gen_sumtree, gen_per_nstep, gen_per_variant_b, gen_noisy,
              gen_dqn_update, gen_sac, gen_normalization]


# --------------------------------------------------------------- H1 trace ----
def gen_ppo_trace(hidden=32, lr=1e-2, name="ppo_trace", slim=False):
    """Row H1: the reference PPOTrainer.train() run unmodified for two
    rollout+update iterations on the build-owned scripted env (tests/scripted_env.py), python /
    numpy / torch seeded 0.  Records every buffer field, the Exp(1) draws Categorical.sample
    consumed, next_value, un-normalised adv/ret, the minibatch index order np.random.shuffle
    produced, the metrics dicts, lr, step_count, episode_rewards and the state_dict after
    each update (ppo_lunarlander.py:198-366)."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    ppo = load_ref("algorithms/ppo_lunarlander.py", "ref_ppo_trace")
    sys.modules["gymnasium"].make = lambda name, **kw: ScriptedEnv(8, 4)
    cfg = ppo.Config()
    cfg.update_freq, cfg.batch_size, cfg.num_epochs = 96, 40, 2                        # 96/40: short last slice
    if hidden is not None:
        cfg.hidden_dim = hidden        # None: the reference's own default (256, ppo_lunarlander.py:43)
    cfg.max_train_steps = 2 * cfg.update_freq
    cfg.lr = lr                        # 1e-2: large enough that ratios clip within two updates
    cfg.device = "cpu"
    seed_all(0)
    tr = ppo.PPOTrainer(cfg)
    out = {"init_" + k: v.numpy().copy() for k, v in tr.model.state_dict().items()}

    noise, rollouts, perms, lrs = [], [], [], []
    orig_get_action = tr.model.get_action

    def get_action(x, deterministic=False):
        st = torch.get_rng_state()
        res = orig_get_action(x, deterministic)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        q = torch.empty(1, 4).exponential_(1.0)
        torch.set_rng_state(after)
        noise.append(q.numpy()[0].copy())
        return res
    tr.model.get_action = get_action

    orig_shuffle = np.random.shuffle

    def shuffle(a):
        orig_shuffle(a)
        perms.append(np.array(a, np.int32))
    np.random.shuffle = shuffle

    orig_update = tr.update
    upd = []

    def update(next_value):
        b = tr.buffer
        adv, ret = tr.compute_gae(next_value)
        rollouts.append(dict(states=np.array(b.states, np.float32), actions=np.array(b.actions, np.int32),
                             log_probs=np.array(b.log_probs, np.float32), values=np.array(b.values, np.float32),
                             rewards=np.array(b.rewards, np.float64), dones=np.array(b.dones, np.uint8),
                             next_value=np.float64(next_value), adv=np.asarray(adv, np.float64),
                             ret=np.asarray(ret, np.float64)))
        lrs.append(tr.optimizer.param_groups[0]["lr"])
        m = orig_update(next_value)
        sd = {k: v.numpy().copy() for k, v in tr.model.state_dict().items()}
        if slim and len(upd) == 0:     # 200 k parameters per snapshot: the first update keeps every 8th element of each
            sd = {k: v.reshape(-1)[::8].copy() for k, v in sd.items()}      # tensor (the second keeps all of them)
        upd.append(dict(metrics=np.array([m["policy_loss"], m["value_loss"], m["entropy"], m["clip_frac"],
                                          m["approx_kl"]], np.float64),
                        step_count=np.int64(tr.step_count), episode_rewards=np.array(tr.episode_rewards, np.float64),
                        **{"sd_" + k: v for k, v in sd.items()}))
        return m
    tr.update = update
    try:
        tr.train()
    finally:
        np.random.shuffle = orig_shuffle
    assert len(rollouts) == 2 and len(perms) == 2 * cfg.num_epochs
    # P8: eval() continues on the training env (:368-399): deterministic argmax episodes
    tr.model.get_action = orig_get_action
    out["eval_episode0"] = np.int64(tr.env.episode + 1)
    out["eval_returns"] = np.array(tr.eval(num_episodes=3), np.float64)
    T = cfg.update_freq
    out["noise_exp"] = np.stack(noise).reshape(2, T, 1, 4).astype(np.float32)
    out["perms"] = np.stack(perms).reshape(2, cfg.num_epochs, T)
    out["lr"] = np.array(lrs, np.float64)
    for r in range(2):
        for k, v in rollouts[r].items():
            out[f"r{r}_{k}"] = v
        for k, v in upd[r].items():
            out[f"r{r}_{k}"] = v
    out["cfg"] = np.array([cfg.update_freq, cfg.batch_size, cfg.num_epochs, cfg.hidden_dim, cfg.max_train_steps], np.int64)
    out["lr0"] = np.float64(cfg.lr)
    out["slim_stride"] = np.int64(8 if slim else 1)
    save(name, **out)


def gen_ppo_trace_h256():
    """The same harness at the reference's OWN network width (hidden_dim 256, the default of ppo_lunarlander.py:43):
    this is the shape gymrl_amd's default update path (hand-written f32-MFMA GEMMs + the loss inside the heads pass,
    ppo_net.FusedActorCriticUpdate.step) runs at, so the replay goes through exactly the kernels bench.py times."""
    gen_ppo_trace(hidden=None, lr=1e-3, name="ppo_trace_h256", slim=True)


def gen_rainbow_update():
    """R4: one full RainbowDQNTrainer.update() (rainbow_dqn_cartpole.py:311-361) — PER sample with the
    uniforms numpy consumed, double-DQN target with fresh NoisyNet noise on both policy forwards (raw
    randn draws recorded), IS-weighted loss, priority update before backward, clip_grad_norm_(10) +
    Adam, soft target update, lr schedule."""
    rb = load_ref("algorithms/rainbow_dqn_cartpole.py", "ref_rainbow_upd")
    cfg = rb.Config()
    cfg.device, cfg.batch_size, cfg.hidden_dim, cfg.memory_capacity = "cpu", 32, 32, 64
    seed_all(90)
    tr = rb.RainbowDQNTrainer(cfg)
    rng = np.random.default_rng(90)
    T = 60
    obs = rng.normal(size=(T + 1, 4)).astype(np.float32)
    act = rng.integers(0, 2, size=T)
    rew = rng.normal(size=T).astype(np.float32)
    done = rng.random(T) < 0.15
    term = done & (rng.random(T) < 0.7)
    for t in range(T):
        tr.memory.store_transition(obs[t], int(act[t]), float(rew[t]), obs[t + 1], bool(term[t]), bool(done[t]))
    with torch.no_grad():
        for p in tr.target_net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    tr.total_steps = 137
    out = {"p0_" + k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()}
    out.update({"t0_" + k: v.numpy().copy() for k, v in tr.target_net.state_dict().items()})
    raw = []

    def scale_noise(size):
        x = torch.randn(size)
        raw.append(x.numpy().copy())
        return x.sign().mul(x.abs().sqrt())
    rb.NoisyLinear.scale_noise = staticmethod(scale_noise)
    np.random.seed(4242)
    loss = tr.update()
    np.random.seed(4242)
    u = np.random.random_sample(cfg.batch_size)
    assert len(raw) == 8, len(raw)           # 2 training-mode forwards x 2 NoisyLinear x (eps_in, eps_out)
    for i, r in enumerate(raw):
        out[f"raw{i}"] = r.astype(np.float32)
    out.update({"p1_" + k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()})
    out.update({"t1_" + k: v.numpy().copy() for k, v in tr.target_net.state_dict().items()})
    out.update(obs=obs, act=act.astype(np.int32), rew=rew, done=done.astype(np.uint8), term=term.astype(np.uint8),
               u=u, loss=np.float64(loss), lr_now=np.float64(tr.optimizer.param_groups[0]["lr"]),
               tree_after=tr.memory.sum_tree.tree.copy(), total_steps=np.int64(137),
               max_episodes=np.int64(cfg.max_episodes), lr0=np.float64(cfg.lr))
    save("rainbow_update", **out)


def gen_buffer_v2():
    """U3: ReplayBuffer_on_policy_v2 (utils/buffer.py:53-102): episodes of different lengths stored row by
    row, sample() trimmed to the longest; all nine returned tensors."""
    sys.path.insert(0, REF)
    from utils.buffer import ReplayBuffer_on_policy_v2
    cfg = types.SimpleNamespace(batch_size=5, max_steps=12, state_shape=(3,), device="cpu")
    buf = ReplayBuffer_on_policy_v2(cfg)
    rng = np.random.default_rng(61)
    lens = [4, 9, 1, 7]                       # the fifth row stays empty (dw == 1 there)
    stream = []
    for L in lens:
        for t in range(L):
            tr = (rng.normal(size=3).astype(np.float32), int(rng.integers(0, 4)), float(np.float32(rng.normal())),
                  float(t == L - 1), float((t == L - 1) and L != 9), float(np.float32(rng.normal())),
                  float(np.float32(rng.normal())), float(np.float32(rng.normal())))
            buf.store(tr)
            stream.append(tr)
        buf.next_episode()
    out = buf.sample()
    names = ["s", "a", "a_logprob", "r", "d", "dw", "v", "v_", "active"]
    res = {"out_" + n: t.numpy() for n, t in zip(names, out)}
    res.update(lens=np.array(lens), stream_s=np.stack([t[0] for t in stream]),
               stream_rest=np.array([t[1:] for t in stream], np.float64), episode_num=np.int64(buf.episode_num))
    save("buffer_v2", **res)


def gen_dqn_trace():
    """H1 for the off-policy loop (rows D3/D5): the reference DQNTrainer.train() run unmodified on the
    scripted env — epsilon schedule advanced per non-deterministic action, python-random exploration,
    push of the pre-reset next_state, update every step once the buffer holds a batch, hard target copy
    every 4 episodes, episode bookkeeping.  Records every python-random draw the loop consumed."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    dq = load_ref("algorithms/dqn_cartpole.py", "ref_dqn_trace")
    env = ScriptedEnv(4, 2)
    env.action_space.sample = lambda: random.randrange(2)
    sys.modules["gymnasium"].make = lambda name, **kw: env
    cfg = dq.Config()
    cfg.device, cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity = "cpu", 32, 16, 400
    cfg.max_episodes, cfg.epsilon_decay, cfg.target_update_freq = 12, 60, 4
    seed_all(123)
    tr = dq.DQNTrainer(cfg)
    with torch.no_grad():                       # wide Q gaps: an argmax cannot flip on 1e-6 differences
        tr.policy_net.net[4].weight.mul_(60.0)
    tr.target_net.load_state_dict(tr.policy_net.state_dict())
    out = {"p0_" + k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()}

    u0, explore_a, idx_log, actions, losses = [], [], [], [], []
    orig_random, orig_sample, orig_randrange = random.random, random.sample, random.randrange

    def rec_random():
        v = orig_random()
        u0.append(v)
        explore_a.append(-1)
        return v

    def rec_randrange(n):
        v = orig_randrange(n)
        explore_a[-1] = v
        return v

    def rec_sample(population, k):
        idx = orig_sample(range(len(population)), k)
        idx_log.append(np.array(idx, np.int32))
        return [population[i] for i in idx]
    random.random, random.sample, random.randrange = rec_random, rec_sample, rec_randrange
    orig_update, orig_select = tr.update, tr.select_action

    def update():
        v = orig_update()
        losses.append(v)
        return v

    def select_action(state, deterministic=False):
        a = orig_select(state, deterministic)
        actions.append(a)
        return a
    tr.update, tr.select_action = update, select_action
    try:
        tr.train()
    finally:
        random.random, random.sample, random.randrange = orig_random, orig_sample, orig_randrange
    out.update({"p1_" + k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()})
    out.update({"t1_" + k: v.numpy().copy() for k, v in tr.target_net.state_dict().items()})
    out.update(u0=np.array(u0, np.float64), explore_a=np.array(explore_a, np.int32), actions=np.array(actions, np.int32),
               indices=np.stack(idx_log), losses=np.array(losses, np.float64),
               episode_rewards=np.array(tr.episode_rewards, np.float64), sample_count=np.int64(tr.sample_count),
               epsilon=np.float64(tr.epsilon),
               cfg=np.array([cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.max_episodes, cfg.epsilon_decay,
                             cfg.target_update_freq], np.int64))
    save("dqn_trace", **out)


def gen_sac_trace():
    """H1 for SAC (row A5): the reference SACTrainer.train() run unmodified on the continuous scripted env —
    stochastic action per step, push, update every step once the buffer holds a batch (critic, actor,
    alpha, soft update), done = terminated or truncated.  Records every N(0,1) draw rsample consumed and
    every replay index."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    sac = load_ref("algorithms/sac_pendulum.py", "ref_sac_trace")
    sys.modules["gymnasium"].make = lambda name, **kw: ScriptedEnv(3, continuous=True)
    cfg = sac.Config()
    cfg.device, cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.max_episodes = "cpu", 32, 16, 400, 8
    seed_all(321)
    tr = sac.SACTrainer(cfg)
    out = {}
    for name, net in (("actor", tr.actor), ("critic", tr.critic)):
        for k, v in net.state_dict().items():
            out[f"p0_{name}_{k}"] = v.numpy().copy()
    act_eps, upd_eps, idx_log, actions, losses = [], [], [], [], []

    def draws(fn, shape_of):
        def wrapped(state, *a, **kw):
            before = torch.get_rng_state()
            res = fn(state, *a, **kw)
            after = torch.get_rng_state()
            torch.set_rng_state(before)
            e = torch.randn(state.shape[0], 1)
            torch.set_rng_state(after)
            shape_of.append(e.numpy().copy())
            return res
        return wrapped
    tr.actor.get_action = draws(tr.actor.get_action, act_eps)
    tr.actor.sample = draws(tr.actor.sample, upd_eps)
    orig_sample = random.sample

    def rec_sample(population, k):
        idx = orig_sample(range(len(population)), k)
        idx_log.append(np.array(idx, np.int32))
        return [population[i] for i in idx]
    random.sample = rec_sample
    orig_update, orig_select = tr.update, tr.select_action

    def update():
        v = orig_update()
        losses.append(v)
        return v

    def select_action(state, deterministic=False):
        a = orig_select(state, deterministic)
        actions.append(np.array(a, np.float32))
        return a
    tr.update, tr.select_action = update, select_action
    try:
        tr.train()
    finally:
        random.sample = orig_sample
    for name, net in (("actor", tr.actor), ("critic", tr.critic), ("critic_target", tr.critic_target)):
        for k, v in net.state_dict().items():
            out[f"p1_{name}_{k}"] = v.numpy().copy()
    n_upd = len(idx_log)
    assert len(upd_eps) == 2 * n_upd
    out.update(act_eps=np.stack(act_eps), eps_next=np.stack(upd_eps[0::2]), eps_cur=np.stack(upd_eps[1::2]),
               indices=np.stack(idx_log), actions=np.stack(actions), losses=np.array(losses, np.float64),
               episode_rewards=np.array(tr.episode_rewards, np.float64), log_alpha=np.float64(tr.log_alpha.item()),
               cfg=np.array([cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.max_episodes], np.int64))
    save("sac_trace", **out)


def gen_rainbow_trace():
    """H1 for Rainbow (row R5): the reference RainbowDQNTrainer.train() run unmodified on the scripted env with
    a 14-step time limit — greedy action on the noisy Q (fresh noise per forward), total_steps, `terminal =
    done and step != max_steps_per_episode - 1` (:376), n-step PER store, update every step once the buffer
    holds a batch (ring of 64 wraps twice).  Records every raw NoisyNet draw and every PER uniform."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    rb = load_ref("algorithms/rainbow_dqn_cartpole.py", "ref_rainbow_trace")
    env = ScriptedEnv(4, 2)
    env.spec.max_episode_steps = 14
    sys.modules["gymnasium"].make = lambda name, **kw: env
    cfg = rb.Config()
    cfg.device, cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.max_episodes = "cpu", 32, 16, 64, 10
    seed_all(777)
    tr = rb.RainbowDQNTrainer(cfg)
    out = {"p0_" + k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()}
    raw = {"advantage": ([], []), "value": ([], [])}
    which = {"layer": None, "k": 0}
    orig_reset = rb.NoisyLinear.reset_noise

    def reset_noise(self):
        which["layer"] = "advantage" if self is tr.policy_net.advantage else ("value" if self is tr.policy_net.value else None)
        which["k"] = 0
        orig_reset(self)
    rb.NoisyLinear.reset_noise = reset_noise

    def scale_noise(size):
        x = torch.randn(size)
        if which["layer"] is not None:
            raw[which["layer"]][which["k"]].append(x.numpy().copy())
            which["k"] += 1
        return x.sign().mul(x.abs().sqrt())
    rb.NoisyLinear.scale_noise = staticmethod(scale_noise)
    actions, losses, us, terminals = [], [], [], []
    orig_update, orig_select, orig_store = tr.update, tr.select_action, tr.memory.store_transition

    def update():
        st = np.random.get_state()
        v = orig_update()
        if len(tr.memory) >= cfg.batch_size:
            st2 = np.random.get_state()
            np.random.set_state(st)
            us.append(np.random.random_sample(cfg.batch_size))
            np.random.set_state(st2)
        losses.append(v)
        return v

    def select_action(state, deterministic=False):
        a = orig_select(state, deterministic)
        actions.append(a)
        return a

    def store_transition(s_, a_, r_, s2_, terminal, done):
        terminals.append(bool(terminal))
        return orig_store(s_, a_, r_, s2_, terminal, done)
    tr.update, tr.select_action, tr.memory.store_transition = update, select_action, store_transition
    tr.train()
    out.update({"p1_" + k: v.numpy().copy() for k, v in tr.policy_net.state_dict().items()})
    out.update({"t1_" + k: v.numpy().copy() for k, v in tr.target_net.state_dict().items()})
    out.update(adv_in=np.stack(raw["advantage"][0]), adv_out=np.stack(raw["advantage"][1]),
               val_in=np.stack(raw["value"][0]), val_out=np.stack(raw["value"][1]), u=np.stack(us),
               actions=np.array(actions, np.int32), losses=np.array(losses, np.float64),
               terminals=np.array(terminals, np.uint8), episode_rewards=np.array(tr.episode_rewards, np.float64),
               total_steps=np.int64(tr.total_steps), tree=tr.memory.sum_tree.tree.copy(),
               lr_now=np.float64(tr.optimizer.param_groups[0]["lr"]),
               cfg=np.array([cfg.hidden_dim, cfg.batch_size, cfg.memory_capacity, cfg.max_episodes, 14], np.int64))
    save("rainbow_trace", **out)


def gen_td3_ddpg():
    """one DDPGTrainer.update() (ddpg_pendulum.py:150-194), two consecutive TD3Trainer.update()
    calls (td3_pendulum.py:171-228: the second one runs the delayed actor + target updates), and the
    exploration noise of both select_action()s, on memories whose whole content is the batch."""
    out = {}
    rng = np.random.default_rng(95)
    for name, path, cls in (("ddpg", "algorithms/ddpg_pendulum.py", "DDPGTrainer"), ("td3", "algorithms/td3_pendulum.py", "TD3Trainer")):
        mod = load_ref(path, "ref_" + name)
        cfg = mod.Config()
        cfg.device, cfg.batch_size, cfg.hidden_dim = "cpu", 24, 32
        seed_all(96)
        tr = getattr(mod, cls)(cfg)
        trans = []
        for i in range(cfg.batch_size):
            trans.append((rng.normal(size=3).astype(np.float32), rng.uniform(-2, 2, size=1),       # actions are float64 arrays
                          float(rng.normal()), rng.normal(size=3).astype(np.float32), bool(rng.random() < 0.15)))
            tr.memory.push(*trans[-1])
        with torch.no_grad():
            for net in (tr.actor_target, tr.critic_target):
                for p in net.parameters():
                    p.add_(0.05 * torch.randn_like(p))
        nets = (("actor", "actor"), ("critic", "critic"), ("actor_target", "actor_target"), ("critic_target", "critic_target"))
        for key, attr in nets:
            for k, v in getattr(tr, attr).state_dict().items():
                out[f"{name}_u0_{key}_{k}"] = v.numpy().copy()
        # exploration: action = clip(actor(s) + N(0, std*bound)) with numpy's global generator
        st = rng.normal(size=3).astype(np.float32)
        np.random.seed(31)
        a = tr.select_action(st)
        np.random.seed(31)
        e = np.random.standard_normal(1)
        out.update({f"{name}_sel_state": st, f"{name}_sel_action": np.asarray(a, np.float64), f"{name}_sel_eps": e,
                    f"{name}_sel_det": np.asarray(tr.select_action(st, deterministic=True), np.float64)})
        losses, orders, epss = [], [], []
        for k in range(1 if name == "ddpg" else 2):
            random.seed(40 + k)
            torch.manual_seed(50 + k)
            losses.append(tr.update())
            random.seed(40 + k)
            orders.append(random.sample(range(cfg.batch_size), cfg.batch_size))
            torch.manual_seed(50 + k)
            epss.append(torch.randn(cfg.batch_size, 1).numpy().astype(np.float64))
        for key, attr in nets:
            for k, v in getattr(tr, attr).state_dict().items():
                out[f"{name}_u1_{key}_{k}"] = v.numpy().copy()
        out.update({f"{name}_orders": np.array(orders, np.int32), f"{name}_eps": np.stack(epss),
                    f"{name}_losses": np.array(losses, np.float64),
                    f"{name}_states": np.stack([t[0] for t in trans]), f"{name}_actions": np.stack([t[1] for t in trans]),
                    f"{name}_rewards": np.array([t[2] for t in trans], np.float32),
                    f"{name}_next_states": np.stack([t[3] for t in trans]),
                    f"{name}_dones": np.array([t[4] for t in trans], np.uint8)})
    save("td3_ddpg", **out)


def gen_dsac():
    """two consecutive SACTrainer.update() calls of the discrete SAC (sac_cartpole.py:148-227) on a
    memory whose whole content is the batch, plus the intermediate tensors of the first call (recomputed here
    from the reference's own networks with the reference's expressions) for the oracle's kernel restatements."""
    mod = load_ref("algorithms/sac_cartpole.py", "ref_dsac")
    cfg = mod.Config()
    cfg.device, cfg.batch_size, cfg.hidden_dim = "cpu", 32, 32
    seed_all(97)
    tr = mod.SACTrainer(cfg)
    rng = np.random.default_rng(98)
    trans = []
    for i in range(cfg.batch_size):
        trans.append((rng.normal(size=4).astype(np.float32), int(rng.integers(0, 2)), float(rng.normal()),
                      rng.normal(size=4).astype(np.float32), bool(rng.random() < 0.15)))
        tr.memory.push(*trans[-1])
    with torch.no_grad():
        for net in (tr.critic1_target, tr.critic2_target):
            for p in net.parameters():
                p.add_(0.05 * torch.randn_like(p))
        tr.log_alpha.fill_(float(np.log(0.2)))          # a temperature at which the entropy terms are visible in f32
    nets = ("actor", "critic1", "critic2", "critic1_target", "critic2_target")
    out = {"states": np.stack([t[0] for t in trans]), "actions": np.array([t[1] for t in trans], np.int32),
           "rewards": np.array([t[2] for t in trans], np.float32), "next_states": np.stack([t[3] for t in trans]),
           "dones": np.array([t[4] for t in trans], np.uint8), "log_alpha0": tr.log_alpha.detach().numpy().copy().reshape(1),
           "gamma": np.float32(cfg.gamma), "target_entropy": np.float32(cfg.target_entropy)}
    for key in nets:
        for k, v in getattr(tr, key).state_dict().items():
            out[f"u0_{key}_{k}"] = v.numpy().copy()
    st = rng.normal(size=4).astype(np.float32)
    out.update({"sel_state": st, "sel_det": np.int32(tr.select_action(st, deterministic=True))})
    # intermediates of the first update, in the batch's storage order
    with torch.no_grad():
        S, S2 = torch.tensor(out["states"]), torch.tensor(out["next_states"])
        R, Dn = torch.tensor(out["rewards"]).unsqueeze(1), torch.tensor(out["dones"], dtype=torch.float32).unsqueeze(1)
        alpha = tr.log_alpha.exp()
        npb = tr.actor(S2)
        nent = -torch.sum(npb * torch.log(npb + 1e-8), dim=1, keepdim=True)
        nq1, nq2 = tr.critic1_target(S2), tr.critic2_target(S2)
        y = R + cfg.gamma * (1 - Dn) * (torch.sum(npb * torch.min(nq1, nq2), dim=1, keepdim=True) + alpha * nent)
        out.update({"k_next_probs": npb.numpy(), "k_next_q1": nq1.numpy(), "k_next_q2": nq2.numpy(), "k_target_q": y.numpy().reshape(-1),
                    "k_q1": tr.critic1(S).numpy(), "k_q2": tr.critic2(S).numpy()})
    q1 = torch.tensor(out["k_q1"], requires_grad=True)
    q2 = torch.tensor(out["k_q2"], requires_grad=True)
    A = torch.tensor(out["actions"], dtype=torch.long).unsqueeze(1)
    l1, l2 = F.mse_loss(q1.gather(1, A), y), F.mse_loss(q2.gather(1, A), y)
    (l1 + l2).backward()
    out.update({"k_dq1": q1.grad.numpy(), "k_dq2": q2.grad.numpy(), "k_critic_losses": np.array([l1.item(), l2.item()])})
    pr = tr.actor(S).detach().requires_grad_(True)
    ent = -torch.sum(pr * torch.log(pr + 1e-8), dim=1, keepdim=True)
    la = torch.sum(pr * torch.min(q1.detach(), q2.detach()), dim=1, keepdim=True)
    al = torch.mean(-alpha.detach() * ent - la)
    al.backward()
    out.update({"k_probs": pr.detach().numpy(), "k_dprobs": pr.grad.numpy(), "k_actor_loss": np.float64(al.item()),
                "k_entropy_mean": np.float64(ent.mean().item())})
    losses, orders, las = [], [], []
    for k in range(2):
        random.seed(60 + k)
        losses.append(tr.update())
        random.seed(60 + k)
        orders.append(random.sample(range(cfg.batch_size), cfg.batch_size))
        las.append(float(tr.log_alpha.item()))
    for key in nets:
        for k, v in getattr(tr, key).state_dict().items():
            out[f"u2_{key}_{k}"] = v.numpy().copy()
    out.update({"orders": np.array(orders, np.int32), "losses": np.array(losses, np.float64), "log_alphas": np.array(las, np.float32)})
    save("dsac", **out)


def _small_lstm_ref(mod, hidden=32, head=32, embed=64):
    """The reference's ActorCritic with its hard-coded 512s (ppo_lstm_lunarlander.py:84-95) replaced by small
    widths so that fixtures stay small; forward / get_action / get_value are the reference's own methods."""
    import torch.nn as nn

    class SmallActorCritic(mod.ActorCritic):
        def __init__(self, state_dim, action_dim, config=None):
            nn.Module.__init__(self)
            self.shared = mod.MHCBackbone(input_dim=state_dim, output_dim=config.mhc_dim, rate=config.mhc_rate,
                                          num_layers=config.mhc_layers, max_sk_it=config.mhc_sk_it)
            self.rnn = mod.URNN(input_size=config.mhc_dim, hidden_size=hidden, layer=nn.GRU)
            self.actor = mod.MLP([hidden, head, action_dim], last_std=0.001)
            self.critic = mod.MLP([hidden, head, 1], last_std=1.0)
            self.rnd = mod.RND(input_dim=state_dim, embed_dim=embed)
    return SmallActorCritic


def gen_ppo_lstm_parts():
    """SURVEY 8f.2 pieces: the reference URNN (nn.GRU) forward + gradients on a short window, the RND reward
    arithmetic (:588-590), and the L4 minibatch loss (:716-776: masked means via the reference's masked_mean,
    clipped value loss) on given logits/values, including a minibatch whose entropy-ratio mask is empty."""
    mod = load_ref("algorithms/ppo_lstm_lunarlander.py", "ref_ppo_lstm")
    cfg = mod.Config()
    from torch.distributions import Categorical
    out = {}
    seed_all(41)
    rnn = mod.URNN(input_size=12, hidden_size=16, layer=torch.nn.GRU)
    x = torch.randn(5, 6, 12, requires_grad=True)
    h0 = (0.5 * torch.randn(5, 16)).requires_grad_(True)
    w_out, w_h = torch.randn(5, 6, 16), torch.randn(5, 16)
    ro, hn = rnn(x, h0)
    ((ro * w_out).sum() + (hn * w_h).sum()).backward()
    out.update({"gru_x": x.detach().numpy(), "gru_h0": h0.detach().numpy(), "gru_w_out": w_out.numpy(), "gru_w_h": w_h.numpy(),
                "gru_out": ro.detach().numpy(), "gru_hn": hn.detach().numpy(), "gru_dx": x.grad.numpy(), "gru_dh0": h0.grad.numpy()})
    for k, v in rnn.state_dict().items():
        out["gru_sd_" + k] = v.numpy().copy()
    for k, p_ in rnn.named_parameters():
        out["gru_grad_" + k] = p_.grad.numpy().copy()
    pred, targ = torch.randn(7, 64).numpy(), torch.randn(7, 64).numpy()
    out.update({"rnd_predict": pred, "rnd_target": targ,
                "rnd_reward": np.array([np.mean((pred[i:i + 1] - targ[i:i + 1]) ** 2) for i in range(7)], np.float32)})
    mm = lambda x_, m_: mod.PPOTrainer.masked_mean(None, x_, m_)          # noqa: E731
    for case, (B, ent_shift) in enumerate(((256, 0.4), (1024, 0.15), (64, 3.0))):
        seed_all(50 + case)
        logits = (torch.randn(B, 4) * 1.5).requires_grad_(True)
        values = torch.randn(B, requires_grad=True)
        with torch.no_grad():
            old_logits = logits * (1.0 + (ent_shift if case == 2 else 0.0)) + ent_shift * torch.randn(B, 4) * (case != 2)
            d_old = Categorical(logits=old_logits)
            a_batch = d_old.sample()
            old_lp, old_ent = d_old.log_prob(a_batch), d_old.entropy()
            old_values = values + 0.3 * torch.randn(B)
        adv_batch, ret_batch = torch.randn(B) * 2, torch.randn(B) * 3
        ent_coef = 0.0123
        dist = Categorical(logits=logits)
        new_lp, new_ent = dist.log_prob(a_batch), dist.entropy()
        entropy_ratio = new_ent / (old_ent + 1e-8)
        erc_mask = ((entropy_ratio > (1 - cfg.erc_beta_low)) & (entropy_ratio < (1 + cfg.erc_beta_high))).float()
        ratio = (new_lp - old_lp).exp()
        covs = (new_lp - new_lp.mean()) * (adv_batch - adv_batch.mean())
        corr = torch.ones_like(adv_batch) * erc_mask
        surr1 = ratio.clamp(0.0, cfg.dual_clip) * adv_batch
        surr2 = torch.clamp(ratio, 1 - cfg.clip_eps_min, 1 + cfg.clip_eps_max) * adv_batch
        clip_frac = mm(((ratio < (1 - cfg.clip_eps_min)) | (ratio > (1 + cfg.clip_eps_max))).float(), corr)
        policy_loss = mm(-torch.min(surr1, surr2), corr)
        value_clip = old_values + (values - old_values).clamp(-cfg.clip_eps_min, cfg.clip_eps_max)
        value_loss = 0.5 * mm(torch.max((values - ret_batch).pow(2), (value_clip - ret_batch).pow(2)), corr)
        entropy = mm(dist.entropy(), corr)
        loss = policy_loss + value_loss + ent_coef * -entropy
        if loss.requires_grad:
            loss.backward()
        zero = lambda t_: torch.zeros_like(t_) if t_.grad is None else t_.grad       # noqa: E731
        pre = f"l{case}_"
        out.update({pre + "logits": logits.detach().numpy(), pre + "values": values.detach().numpy(),
                    pre + "actions": a_batch.numpy().astype(np.int32), pre + "old_lp": old_lp.numpy(),
                    pre + "old_ent": old_ent.numpy(), pre + "old_values": old_values.numpy(), pre + "adv": adv_batch.numpy(),
                    pre + "ret": ret_batch.numpy(), pre + "dlogits": zero(logits).numpy(), pre + "dvalues": zero(values).numpy(),
                    pre + "metrics": np.array([float(policy_loss), float(value_loss), float(entropy), float(clip_frac),
                                               (old_lp - new_lp).mean().item(), 1.0 - erc_mask.mean().item(),
                                               covs.mean().item(), erc_mask.sum().item()], np.float64)})
    out["n_cases"] = np.int64(3)
    out["cfg"] = np.array([cfg.clip_eps_min, cfg.clip_eps_max, cfg.dual_clip, cfg.erc_beta_low, cfg.erc_beta_high, 0.0123], np.float64)
    save("ppo_lstm_parts", **out)


def gen_ppo_lstm_trace():
    """Row H1 for the recurrent trainer: the reference PPOTrainer.train() (ppo_lstm_lunarlander.py:814-822) run
    for two collect -> advantages -> update iterations on the scripted env with the small-width ActorCritic.
    Records the Exp(1) draws Categorical.sample consumed, the sequence permutations torch.randperm produced,
    every buffer field incl. the stored hidden states and the RND-augmented rewards, adv/returns, per-minibatch
    gradient norms and mask counts, lr / ent_coef / step_count / episode_rewards and the weights after each update."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    mod = load_ref("algorithms/ppo_lstm_lunarlander.py", "ref_ppo_lstm_trace")
    sys.modules["gymnasium"].make = lambda name, **kw: ScriptedEnv(8, 4)
    mod.ActorCritic = _small_lstm_ref(mod)
    cfg = mod.Config()
    cfg.update_freq, cfg.seq_len, cfg.batch_size, cfg.num_epochs = 64, 8, 4, 2
    cfg.mhc_dim, cfg.mhc_layers, cfg.mhc_sk_it = 16, 1, 4
    cfg.max_train_steps, cfg.lr, cfg.seed = 2 * cfg.update_freq, 5e-3, 3
    seed_all(0)
    tr = mod.PPOTrainer(cfg)
    with torch.no_grad():                                   # away from the near-uniform init policy (last_std = 0.001)
        for n_, p_ in tr.model.named_parameters():
            if n_.startswith("actor") or n_.endswith(".w") or n_.endswith(".alpha"):
                p_.add_(0.25 * torch.randn_like(p_))
    out = {"init_" + k: v.numpy().copy() for k, v in tr.model.state_dict().items()}
    noise, perms, gnorms, counts, rollouts, upd = [], [], [], [], [], []
    orig_get_action = tr.model.get_action

    def get_action(x, hidden_state, deterministic=False):
        st = torch.get_rng_state()
        res = orig_get_action(x, hidden_state, deterministic)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        q = torch.empty(1, 4).exponential_(1.0)
        torch.set_rng_state(after)
        noise.append(q.numpy()[0].copy())
        return res
    tr.model.get_action = get_action
    orig_randperm, orig_clip, orig_mm = torch.randperm, mod.nn.utils.clip_grad_norm_, tr.masked_mean

    def randperm(n, *a, **k):
        r = orig_randperm(n, *a, **k)
        if n == tr.num_sequences:
            perms.append(r.numpy().astype(np.int32))
        return r

    def clip(params, max_norm, *a, **k):
        tn = orig_clip(params, max_norm, *a, **k)
        gnorms.append(float(tn))
        return tn

    def masked_mean(x, mask=None):
        counts.append(float(mask.sum()))
        return orig_mm(x, mask)
    orig_update = tr.update_model

    def update_model(adv, ret):
        b = tr.buffer
        rollouts.append(dict(states=np.array(b.states, np.float32), actions=np.array(b.actions, np.int32),
                             log_probs=np.array(b.log_probs, np.float32), values=np.array(b.values, np.float32),
                             rewards=np.array(b.rewards, np.float64), dones=np.array(b.dones, np.uint8),
                             old_entropies=np.array(b.old_entropies, np.float32),
                             hidden_states=np.array(b.hidden_states, np.float32), next_value=np.float64(b.next_value),
                             adv=np.asarray(adv, np.float64), ret=np.asarray(ret, np.float64)))
        orig_update(adv, ret)
        upd.append(dict(lr=np.float64(tr.lr), ent_coef=np.float64(tr.ent_coef), step_count=np.int64(tr.step_count),
                        episode_rewards=np.array(tr.episode_rewards, np.float64),
                        **{"sd_" + k: v.numpy().copy() for k, v in tr.model.state_dict().items()}))
    tr.update_model = update_model
    tr.masked_mean = masked_mean
    torch.randperm, mod.nn.utils.clip_grad_norm_ = randperm, clip
    try:
        tr.train()
    finally:
        torch.randperm, mod.nn.utils.clip_grad_norm_ = orig_randperm, orig_clip
    T, n_mb = cfg.update_freq, tr.num_sequences // cfg.batch_size
    assert len(rollouts) == 2 and len(perms) == 2 * cfg.num_epochs and len(gnorms) == 2 * cfg.num_epochs * n_mb
    out["noise_exp"] = np.stack(noise).reshape(2, T, 1, 4).astype(np.float32)
    out["perms"] = np.stack(perms).reshape(2, cfg.num_epochs, tr.num_sequences)
    out["grad_norms"] = np.array(gnorms, np.float64).reshape(2, cfg.num_epochs * n_mb)
    out["mask_counts"] = np.array(counts, np.float64).reshape(2, cfg.num_epochs * n_mb, 4)[:, :, 0]
    for r in range(2):
        for k, v in rollouts[r].items():
            out[f"r{r}_{k}"] = v
        for k, v in upd[r].items():
            out[f"r{r}_{k}"] = v
    out["cfg"] = np.array([cfg.update_freq, cfg.seq_len, cfg.batch_size, cfg.num_epochs, cfg.mhc_dim, cfg.mhc_layers,
                           cfg.mhc_sk_it, cfg.max_train_steps, cfg.seed], np.int64)
    out["lr0"] = np.float64(cfg.lr)
    save("ppo_lstm_trace", **out)


def gen_ppo_full_trace():
    """Row F3 / H1 for PPO-full: the reference PPOTrainer.train() (ppo_full_lunarlander.py:681-700) run unmodified for two
    collect_experience -> compute_advantages -> update_model iterations on the scripted env at small mHC widths.
    Records the Exp(1) draws Categorical.sample consumed, the DataLoader's shuffle orders (RandomSampler's torch.randperm),
    every buffer field incl. old_entropies, next_value, adv / returns, per-minibatch gradient norms, lr / ent_coef
    (annealed AFTER the update, :660-666), step_count, episode_rewards and the weights after each update."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    mod = load_ref("algorithms/ppo_full_lunarlander.py", "ref_ppo_full_trace")
    sys.modules["gymnasium"].make = lambda name, **kw: ScriptedEnv(8, 4)
    cfg = mod.Config()
    cfg.update_freq, cfg.batch_size, cfg.num_epochs = 96, 40, 2          # 96 / 40: a short last minibatch
    cfg.mhc_dim, cfg.mhc_layers, cfg.mhc_sk_it = 16, 1, 4
    cfg.max_train_steps, cfg.lr, cfg.seed = 2 * cfg.update_freq, 5e-3, 3
    seed_all(0)
    tr = mod.PPOTrainer(cfg)
    with torch.no_grad():                                   # away from the near-uniform init policy (last_std = 0.001)
        for n_, p_ in tr.model.named_parameters():
            if n_.startswith("actor") or n_.endswith(".w") or n_.endswith(".alpha"):
                p_.add_(0.25 * torch.randn_like(p_))
    out = {"init_" + k: v.numpy().copy() for k, v in tr.model.state_dict().items()}
    noise, perms, gnorms, rollouts, upd = [], [], [], [], []
    orig_get_action = tr.model.get_action

    def get_action(x, deterministic=False):
        st = torch.get_rng_state()
        res = orig_get_action(x, deterministic)
        after = torch.get_rng_state()
        torch.set_rng_state(st)
        q = torch.empty(1, 4).exponential_(1.0)
        torch.set_rng_state(after)
        noise.append(q.numpy()[0].copy())
        return res
    tr.model.get_action = get_action
    orig_randperm, orig_clip = torch.randperm, mod.nn.utils.clip_grad_norm_

    def randperm(n, *a, **k):
        r = orig_randperm(n, *a, **k)
        if n == cfg.update_freq:
            perms.append(r.numpy().astype(np.int32))
        return r

    def clip(params, max_norm, *a, **k):
        tn = orig_clip(params, max_norm, *a, **k)
        gnorms.append(float(tn))
        return tn
    orig_update = tr.update_model

    def update_model(adv, ret):
        b = tr.buffer
        rollouts.append(dict(states=np.array(b.states, np.float32), actions=np.array(b.actions, np.int32),
                             log_probs=np.array(b.log_probs, np.float32), values=np.array(b.values, np.float32),
                             rewards=np.array(b.rewards, np.float64), dones=np.array(b.dones, np.uint8),
                             old_entropies=np.array(b.old_entropies, np.float32), next_value=np.float64(b.next_value),
                             adv=np.asarray(adv, np.float64), ret=np.asarray(ret, np.float64)))
        orig_update(adv, ret)
        upd.append(dict(lr=np.float64(tr.lr), ent_coef=np.float64(tr.ent_coef), step_count=np.int64(tr.step_count),
                        episode_rewards=np.array(tr.episode_rewards, np.float64),
                        **{"sd_" + k: v.numpy().copy() for k, v in tr.model.state_dict().items()}))
    tr.update_model = update_model
    torch.randperm, mod.nn.utils.clip_grad_norm_ = randperm, clip
    try:
        tr.train()
    finally:
        torch.randperm, mod.nn.utils.clip_grad_norm_ = orig_randperm, orig_clip
    T = cfg.update_freq
    n_mb = (T + cfg.batch_size - 1) // cfg.batch_size
    # RandomSampler.__iter__ draws a second, unused randperm(n) at the end of every epoch (its `num_samples % n` tail)
    assert (len(rollouts), len(perms), len(gnorms)) == (2, 4 * cfg.num_epochs, 2 * cfg.num_epochs * n_mb), \
        (len(rollouts), len(perms), len(gnorms))
    perms = perms[::2]
    out["noise_exp"] = np.stack(noise).reshape(2, T, 1, 4).astype(np.float32)
    out["perms"] = np.stack(perms).reshape(2, cfg.num_epochs, T)
    out["grad_norms"] = np.array(gnorms, np.float64).reshape(2, cfg.num_epochs * n_mb)
    for r in range(2):
        for k, v in rollouts[r].items():
            out[f"r{r}_{k}"] = v
        for k, v in upd[r].items():
            out[f"r{r}_{k}"] = v
    out["cfg"] = np.array([cfg.update_freq, cfg.batch_size, cfg.num_epochs, cfg.mhc_dim, cfg.mhc_layers, cfg.mhc_sk_it,
                           cfg.max_train_steps, cfg.seed], np.int64)
    out["lr0"] = np.float64(cfg.lr)
    save("ppo_full_trace", **out)


def gen_runner_trace():
    """Row H1/U1: the reference utils/runner.py `train()` run unmodified on the scripted env with toy
    agents written to the legacy duck-type (outputs are a function of the call index only, so that they are exact
    on both sides): the on-policy branch (choose-next-action-before-store, V(next) of the terminal observation,
    :119-131) and the off-policy branch (store, choose, then update, :132-140).  Records every stored transition
    (normalised states, scaled rewards), the order of choose_action / update / save_model calls, and the final
    running statistics."""
    sys.path.insert(0, os.path.dirname(OUT))
    from scripted_env import ScriptedEnv
    _install_stub_gym()
    g = sys.modules["gymnasium"]

    class _Box:
        pass
    g.spaces = types.SimpleNamespace(Box=_Box)
    g.ObservationWrapper = object
    wrappers = types.ModuleType("gymnasium.wrappers")
    wrappers.AtariPreprocessing = object
    sys.modules["gymnasium.wrappers"] = wrappers
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            if name == "catch":
                return lambda *a, **k: (lambda f: f)
            return lambda *a, **k: None
    lg.logger = _Logger()
    sys.modules["loguru"] = lg
    sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[m]
    import utils.runner as rr                                   # the reference's runner, unmodified
    out = {}
    for kind in ("on", "off"):
        env = ScriptedEnv(8, 4)
        cfg = rr.BasicConfig()
        cfg.env_name, cfg.algo_name, cfg.train_eps, cfg.eval_freq, cfg.save_freq = "Scripted", "toy", 7, 10 ** 9, 3
        cfg.max_steps, cfg.batch_size, cfg.gamma, cfg.lamda, cfg.device = 500, 20, 0.99, 0.95, "cpu"
        cfg.memory_capacity = 10 ** 6
        calls, stored = [], []

        class Toy:
            def __init__(self):
                self.cfg, self.k, self.learn_step = cfg, 0, 0
                self.net = object()
                self.memory = rr.ReplayBuffer_on_policy(cfg) if kind == "on" else rr.ReplayBuffer_off_policy(cfg)

            def choose_action(self, state):
                self.k += 1
                calls.append(1)
                a = (self.k * 5 + 1) % 4
                return (a, -0.125 * self.k, 0.25 * self.k) if kind == "on" else a

            def update(self):
                calls.append(2)
                if kind == "on":
                    self.memory.clear()
                self.learn_step += 1
                return {}

            def save_model(self):
                calls.append(3)
        agent = Toy()
        store = agent.memory.store

        def rec(tr):
            stored.append(tr)
            store(tr)
        agent.memory.store = rec
        np.random.seed(5)
        rr.train(env, agent, cfg)
        if kind == "on":
            out["on_state"] = np.array([t[0] for t in stored], np.float64)
            for j, name in ((1, "action"), (2, "reward"), (3, "done"), (4, "dw"), (5, "log_prob"), (6, "value"), (7, "next_value")):
                out["on_" + name] = np.array([t[j] for t in stored], np.float64)
        else:
            out["off_state"] = np.array([t[0] for t in stored], np.float64)
            out["off_next_state"] = np.array([t[3] for t in stored], np.float64)
            for j, name in ((1, "action"), (2, "reward"), (4, "done")):
                out["off_" + name] = np.array([t[j] for t in stored], np.float64)
        out[kind + "_calls"] = np.array(calls, np.int8)
        ms, rs = agent.state_norm.running_ms, agent.reward_scaler.running_ms
        out[kind + "_norm"] = np.concatenate([[ms.n], np.asarray(ms.mean, np.float64), np.asarray(ms.std, np.float64)])
        out[kind + "_rscale"] = np.array([rs.n, float(np.asarray(rs.mean).reshape(-1)[0]), float(np.asarray(rs.std).reshape(-1)[0])])
        out[kind + "_learn_step"] = np.int64(agent.learn_step)
    save("runner_trace", **out)


def gen_ppo_full_pscn():
    """ppo_full ActorCritic with use_mhc = False (PSCN trunk, ppo_full_lunarlander.py:377-384): forward of the
    network as initialised under torch seed 77 — the fixture holds no weights, the test rebuilds them from the seed
    (same construction order = same draws) and checks a checksum of them."""
    pf = load_ref("algorithms/ppo_full_lunarlander.py", "ref_ppo_full_pscn")
    cfg = pf.Config()
    cfg.use_mhc = False
    torch.manual_seed(77)
    net = pf.ActorCritic(8, 4, config=cfg)
    x = torch.randn(16, 8)
    logits, values = net(x)
    chk = np.array([float(v.double().sum()) for v in net.state_dict().values()], np.float64)
    save("ppo_full_pscn", x=x.numpy(), logits=logits.detach().numpy(), values=values.detach().numpy(), checksum=chk,
         keys=np.array(list(net.state_dict().keys())))


if __name__ == "__main__":
    names = sys.argv[1:]
    for g in GENERATORS + [gen_ppo_trace, gen_ppo_trace_h256, gen_rainbow_update, gen_buffer_v2, gen_dqn_trace, gen_sac_trace, gen_rainbow_trace, gen_td3_ddpg, gen_dsac, gen_ppo_lstm_parts, gen_ppo_lstm_trace, gen_runner_trace, gen_ppo_full_pscn, gen_ppo_full_trace]:
        if not names or g.__name__ in names:
            g()
