"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" == RCCL on
ROCm, over xGMI inside a node).  Env instances shard naturally (rank r owns env ids
[r*N, (r+1)*N)); the only exchanges on the path are
  * one all-reduce(SUM) of the flat fp32 gradient buffer per optimiser step
    (0.80 MB for the PPO MLP; latency-bound, so it is ONE collective on ONE buffer),
  * one all-reduce of 3 float64 advantage moments per rollout (ppo_lunarlander.py:236
    normalises over the whole rollout),
  * a broadcast of the initial parameters.
The reference has no distributed code at all.
"""
import os

import torch
import torch.distributed as td


def is_dist():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_dist() else 0


def world_size():
    return td.get_world_size() if is_dist() else 1


def init_from_env(backend=None):
    """Initialise from torchrun-style env vars (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world_size, local_rank).  No-op for WORLD_SIZE <= 1."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lrk = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(lrk)
        td.init_process_group(backend=backend, rank=rk, world_size=ws)
    elif torch.cuda.is_available():
        torch.cuda.set_device(lrk)
    return rk, ws, lrk


def all_reduce_sum(t):
    if is_dist() and world_size() > 1:
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_max(t):
    if is_dist() and world_size() > 1:
        td.all_reduce(t, op=td.ReduceOp.MAX)
    return t


def broadcast(t, src=0):
    if is_dist() and world_size() > 1:
        td.broadcast(t, src=src)
    return t


def barrier():
    if is_dist() and world_size() > 1:
        td.barrier()


def shutdown():
    """Collective teardown: every rank waits for the slowest one, then the process group is destroyed
    (ranks that simply exit while another still runs local work make RCCL's watchdog noisy)."""
    if is_dist():
        if world_size() > 1:
            td.barrier()
        td.destroy_process_group()


def shard_env_ids(num_envs_per_rank, rk=None):
    """Global env-id range owned by a rank (the Philox stream is keyed by env id, so
    an N-GPU run steps exactly the env instances a 1-GPU run with N*num_envs would)."""
    rk = rank() if rk is None else rk
    return rk * num_envs_per_rank, (rk + 1) * num_envs_per_rank
