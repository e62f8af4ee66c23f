"""Worker of tests/test_multirank_gpu.py: one of `world` gloo ranks sharing cuda:0 (the 1-GPU lease has no second
device; RCCL needs one device per rank, gloo does not) — or ONE "nccl" (= RCCL) rank under GYMRL_FORCE_COLLECTIVES=1, or no
process group at all ("none": the run the forced one must equal bit for bit).  Runs one iteration of a trainer through
its collective branches and leaves what the parent checks in `outdir`."""
import os
import sys

import torch
import torch.distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def ppo_cfg(n_envs):
    from gymrl_amd.ppo_lunarlander import Config
    cfg = Config()
    cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.num_minibatches, cfg.seed = n_envs, 48, 2, 4, 5
    return cfg


def ppo_full_cfg(n_envs):
    from gymrl_amd.ppo_full_lunarlander import Config
    cfg = Config()
    cfg.num_envs, cfg.update_freq, cfg.num_epochs, cfg.batch_size, cfg.seed, cfg.mhc_dim = n_envs, 32, 2, 256, 1, 32
    return cfg


def run(rank, world, port, outdir, n_envs, backend="gloo"):
    # "nccl" with more than one rank needs a device per rank (the day a lease has two: rank r on cuda:r); everything else shares cuda:0
    local = rank if (backend == "nccl" and world > 1) else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local))
    torch.cuda.set_device(local)
    if backend != "none":
        td.init_process_group(backend, rank=rank, world_size=world)
    try:
        from gymrl_amd import dist as gdist
        from gymrl_amd.ppo_lunarlander import PPOTrainer
        tr = PPOTrainer(ppo_cfg(n_envs))
        assert tr.world_size == world and tr.env.env_id0 == rank * n_envs
        assert tr.collective == (backend != "none") and gdist.backend() == (None if backend == "none" else backend)
        p0 = tr.flat_params.clone()
        nv = tr.collect_rollout()
        from gymrl_amd.ppo_lunarlander import KernelTimers
        tr._timers = KernelTimers()                 # bench.py's mode: the reducer brackets its buckets with HIP events
        m = tr.update(nv)
        tr._timers = None
        b = tr.buffer
        rs = tr._reducer.stats() if tr._reducer is not None else None
        out = dict(backend=gdist.backend(),
                   p0=p0.cpu(), params=tr.flat_params.cpu(), moments=tr._moments.cpu(), adv=b.advantages.cpu(),
                   metrics={k: float(v) for k, v in m.items()}, reducer=rs,
                   **{k: getattr(b, k).cpu() for k in ("states", "actions", "log_probs", "values", "rewards", "dones")})
        # PPO-full: two iterations (the second replays the two hipGraphs around the eager all-reduce)
        from gymrl_amd.ppo_full_lunarlander import PPOTrainer as FullTrainer
        torch.manual_seed(5)
        ft = FullTrainer(ppo_full_cfg(n_envs))
        fm = []
        for _ in range(2):
            ft.collect_experience()
            adv, ret = ft.compute_advantages()
            fm.append(ft.update_model(adv, ret))
        out.update(full_params=ft.flat_params.cpu(), full_metrics=fm, full_graphed=ft._g_idx is not None,
                   full_actions=ft.buffer.actions.cpu(), full_steps=ft.optimizer.step_count)
        torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        if backend != "none":
            td.barrier()
            td.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), *(sys.argv[6:7]))
