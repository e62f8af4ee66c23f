"""N > 1 path on CPU: two gloo processes exercise gymrl_amd/dist.py exactly as the trainers use
it — env-id sharding, the flat-gradient all-reduce with grad_scale = 1/world (ranks end up with
identical parameters), the 3-float64 advantage-moment all-reduce (global normalisation ==
normalisation of the concatenated shards), parameter broadcast, max-over-ranks timing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from gymrl_amd import dist as gdist
    from oracle import oracle as orc
    rk, ws, _ = gdist.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, world) and gdist.world_size() == world
    # env shards: rank r owns global env ids [r*N, (r+1)*N); Philox streams are keyed by them
    N = 8
    lo, hi = gdist.shard_env_ids(N)
    obs = orc.Env(orc.LUNARLANDER, N, seed=3, env_id0=lo).reset()
    # parameter broadcast
    p = torch.full((10,), float(rank + 1))
    gdist.broadcast(p)
    assert torch.all(p == 1.0)
    # flat-gradient all-reduce + identical update on every rank
    torch.manual_seed(100 + rank)
    g = torch.randn(1000)
    g_local = g.clone()
    gdist.all_reduce_sum(g)
    params, _, m, v = orc.adam_step(np.zeros(1000, np.float32), g.numpy(), np.zeros(1000), np.zeros(1000), 3e-4, 0.9,
                                    0.999, 1e-5, 1, grad_scale=1.0 / world, max_grad_norm=0.5)
    # advantage moments: (count, sum, sumsq) summed over ranks
    rng = np.random.default_rng(7 + rank)
    adv = (rng.normal(size=(16, N)) * (1 + rank) + rank).astype(np.float32)
    mom = torch.from_numpy(orc.moments(adv))
    gdist.all_reduce_sum(mom)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    gdist.all_reduce_max(t)
    # bucketed reducer (the trainers' path): two buckets launched back to front, wait(), result == plain all-reduce
    torch.manual_seed(200 + rank)
    gb = torch.randn(1000)
    gb_local = gb.clone()
    red = gdist.GradReducer(gb, [640])
    assert [b.numel() for b in red.buckets] == [640, 360]
    red.launch(1)
    red.launch(0)
    red.wait()
    assert red.stats() is None                       # untimed on CPU tensors
    devs = gdist.rank_devices()
    assert [d["rank"] for d in devs] == list(range(world))
    gdist.barrier()
    ret[rank] = dict(obs=obs, params=params, g_local=g_local.numpy(), mom=mom.numpy(), adv=adv, tmax=float(t.item()),
                     gb=gb.numpy(), gb_local=gb_local.numpy())


@pytest.mark.timeout(300)
def test_two_rank_gloo_data_parallel_path():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    a, b = ret[0], ret[1]
    # identical parameters after the averaged, clipped Adam step
    assert np.array_equal(a["params"], b["params"])
    from oracle import oracle as orc
    mean_g = (a["g_local"] + b["g_local"])
    ref, _, _, _ = orc.adam_step(np.zeros(1000, np.float32), mean_g, np.zeros(1000), np.zeros(1000), 3e-4, 0.9, 0.999,
                                 1e-5, 1, grad_scale=0.5, max_grad_norm=0.5)
    assert np.allclose(a["params"], ref, atol=1e-7)
    # global advantage normalisation == normalisation over the concatenation of both shards
    both = np.concatenate([a["adv"], b["adv"]], axis=1)
    assert np.allclose(a["mom"], orc.moments(both), rtol=1e-12) and np.array_equal(a["mom"], b["mom"])
    # env shards tile the global id space: 2 ranks x 8 envs == 1 rank x 16 envs
    full = orc.Env(orc.LUNARLANDER, 16, seed=3, env_id0=0).reset()
    assert np.array_equal(np.concatenate([a["obs"], b["obs"]]), full)
    assert a["tmax"] == b["tmax"] == pytest.approx(0.2)
    assert np.array_equal(a["gb"], b["gb"]) and np.array_equal(a["gb"], a["gb_local"] + b["gb_local"])


def test_grad_reducer_rejects_bounds_outside_the_buffer():
    sys.path.insert(0, ROOT)
    from gymrl_amd import dist as gdist
    g = torch.zeros(100)
    with pytest.raises(ValueError):
        gdist.GradReducer(g, [200])
    r = gdist.GradReducer(g, [40, 40, 0, 100])
    assert [b.numel() for b in r.buckets] == [40, 60]
    r.launch(0)                                      # world_size 1: no-ops
    r.wait()


def _forced_worker(port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gymrl_amd import dist as gdist
    os.environ["GYMRL_FORCE_COLLECTIVES"] = "0"
    assert gdist.init_from_env(backend="gloo") == (0, 1, 0) and not gdist.is_dist() and not gdist.collectives_active()
    os.environ["GYMRL_FORCE_COLLECTIVES"] = "1"
    assert gdist.init_from_env(backend="gloo") == (0, 1, 0) and gdist.is_dist() and gdist.collectives_active()
    assert gdist.backend() == "gloo" and gdist.world_size() == 1
    g = torch.arange(10, dtype=torch.float32)
    red = gdist.GradReducer(g, [4])
    red.launch(1)
    red.launch(0)
    red.wait()
    t = torch.tensor([3.0], dtype=torch.float64)
    gdist.all_reduce_sum(t)
    gdist.all_reduce_max(t)
    gdist.broadcast(t)
    gdist.barrier()
    ret["ok"] = bool(torch.equal(g, torch.arange(10, dtype=torch.float32)) and t.item() == 3.0
                     and len(gdist.rank_devices()) == 1)
    gdist.shutdown()
    ret["down"] = not gdist.is_dist()


def test_forced_collectives_with_one_rank_are_the_identity():
    """GYMRL_FORCE_COLLECTIVES=1 (dist.force_collectives): ONE rank builds a process group and every helper issues its
    collective — the switch tests/test_multirank_gpu.py uses to take the trainers through RCCL on a 1-GPU box."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        p = ctx.Process(target=_forced_worker, args=(_free_port(), ret))
        p.start()
        p.join(120)
        assert p.exitcode == 0 and ret.get("ok") and ret.get("down")
