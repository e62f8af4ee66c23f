"""Checkpoint / resume in the reference's format.

utils/model.py:337-366 (`ModelLoader.save_model / load_model`) stores one dict per agent:
`<attr>_state_dict` for every attribute that has a `state_dict()` (net, optimizer, ...) and the
plain attributes (learn_step, state_norm, ...) under their own names; `torch.save` / `torch.load`.
The trainers here keep parameters and Adam moments in flat device buffers, so this module

  * writes the network as its ordinary `state_dict()` (same keys as the reference modules) and
    the fused optimiser as a **torch.optim.Adam-format** state dict (per-parameter `step`,
    `exp_avg`, `exp_avg_sq`; one param group) that `torch.optim.Adam.load_state_dict` accepts;
  * reads either back into the flat buffers, so a run can resume bit-exactly, and a checkpoint
    written by a reference-style agent (`net_state_dict` + `optimizer_state_dict`) can be loaded.
"""
import torch


def _offsets(model):
    """(param, offset, numel) of every parameter inside the module's flat buffer."""
    flat = model._flat_params
    return [(p, (p.data_ptr() - flat.data_ptr()) // 4, p.numel()) for p in model.parameters()]


def adam_state_dict(model, opt):
    """FusedAdam -> torch.optim.Adam.state_dict() layout (parameters in model.parameters() order)."""
    state = {}
    if opt.step_count > 0:
        for i, (p, off, n) in enumerate(_offsets(model)):
            state[i] = {"step": torch.tensor(float(opt.step_count)),
                        "exp_avg": opt.m[off:off + n].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": opt.v[off:off + n].view(p.shape).detach().cpu().clone()}
    g = opt.param_groups[0]
    group = {"lr": float(g["lr"]), "betas": tuple(g["betas"]), "eps": float(g["eps"]), "weight_decay": 0,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "decoupled_weight_decay": False, "params": list(range(len(list(model.parameters()))))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(model, opt, sd):
    """torch.optim.Adam-format dict -> FusedAdam's flat moments (all parameters share one step)."""
    g = sd["param_groups"][0]
    opt.param_groups[0].update(lr=float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]))
    steps = set()
    opt.m.zero_()
    opt.v.zero_()
    for i, (p, off, n) in enumerate(_offsets(model)):
        st = sd["state"].get(i)
        if st is None:
            continue
        steps.add(int(float(st["step"])))
        opt.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(opt.m.device, torch.float32))
        opt.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(opt.v.device, torch.float32))
    if len(steps) > 1:
        raise ValueError(f"parameters with different Adam step counts {sorted(steps)} cannot share a fused step")
    opt.step_count = steps.pop() if steps else 0


def save_agent(path, nets, optimizers, **attrs):
    """One ModelLoader-style dict: `<name>_state_dict` for nets / optimisers, attrs verbatim.
    nets: {name: module}; optimizers: {name: (module, FusedAdam)}."""
    state = {f"{k}_state_dict": {n: t.detach().cpu() for n, t in m.state_dict().items()} for k, m in nets.items()}
    for k, (m, opt) in optimizers.items():
        state[f"{k}_state_dict"] = adam_state_dict(m, opt)
    state.update(attrs)
    torch.save(state, path)
    return state


def load_agent(path, nets, optimizers, map_location="cpu"):
    """Inverse of save_agent; returns the remaining plain attributes.  Module parameters are copied
    into their existing (flat-buffer) storage, never re-bound."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    for k, m in nets.items():
        m.load_state_dict(ck[f"{k}_state_dict"])
    for k, (m, opt) in optimizers.items():
        if f"{k}_state_dict" in ck:
            load_adam_state_dict(m, opt, ck[f"{k}_state_dict"])
    skip = {f"{k}_state_dict" for k in list(nets) + list(optimizers)}
    return {k: v for k, v in ck.items() if k not in skip}
