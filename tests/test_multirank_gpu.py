"""Section 8(e): the trainers' world_size > 1 branches, executed.  Two gloo ranks share cuda:0 (one process per rank,
env ids [rN, (r+1)N), flat-gradient all-reduce with grad_scale = 1/world inside clip + Adam, all-reduced advantage
moments, broadcast initial parameters); checked against a single-rank run with 2N envs."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_ppo_and_ppo_full(tmp_path):
    N, world = 64, 2
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GYMRL_FORCE_COLLECTIVES", None)
    # two RCCL ranks on two devices where a lease has them; the 1-GPU boxes of this pool run the same ranks over gloo on cuda:0
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), str(port),
                               str(tmp_path), str(N), backend], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt"), weights_only=False) for k in range(world)]
    assert all(x["backend"] == backend for x in r)
    # identical initial parameters (broadcast) and bit-identical parameters after 8 all-reduced optimiser steps
    assert torch.equal(r[0]["p0"], r[1]["p0"]) and torch.equal(r[0]["params"], r[1]["params"])
    assert not torch.equal(r[0]["params"], r[0]["p0"])
    # the gradient travelled as two buckets per optimiser step (tail of the flat buffer first), on the reducer's stream
    for k in (0, 1):
        rs = r[k]["reducer"]
        assert rs["collectives"] == 2 * 8 and rs["stalls"] == 8 and len(rs["bucket_bytes"]) == 2
        assert rs["bucket_bytes"][1] > rs["bucket_bytes"][0] > 0 and rs["collective_s"] > 0
    assert torch.equal(r[0]["moments"], r[1]["moments"])
    # rank r's slab == envs [rN, (r+1)N) of a single-rank run with 2N envs (same seed, same initial weights)
    sys.path.insert(0, HERE)
    from multirank_worker import ppo_cfg
    from gymrl_amd.ppo_lunarlander import PPOTrainer
    one = PPOTrainer(ppo_cfg(world * N))
    assert torch.equal(one.flat_params.cpu(), r[0]["p0"])
    one.collect_rollout()
    one.compute_gae()
    b = one.buffer
    for k in range(world):
        sl = slice(k * N, (k + 1) * N)
        for name in ("states", "actions", "log_probs", "values", "rewards", "dones"):
            assert torch.equal(getattr(b, name)[:, sl].cpu(), r[k][name]), (k, name)
        assert torch.equal(b.advantages[:, sl].cpu(), r[k]["adv"])
    # all-reduced moments == the moments of the concatenated shards
    m1, m2 = one._moments.cpu().numpy(), r[0]["moments"].numpy()
    assert m1[0] == m2[0] == world * N * b.T and np.allclose(m1[1:], m2[1:], rtol=1e-12, atol=1e-9)
    # different shards -> different per-rank metrics, same weights
    assert r[0]["metrics"] != r[1]["metrics"]
    # PPO-full: graphs kept under world_size > 1, parameters identical on both ranks, both ranks stepped 2 x 2 x 8 times
    assert r[0]["full_graphed"] and r[1]["full_graphed"]
    assert torch.equal(r[0]["full_params"], r[1]["full_params"]) and r[0]["full_steps"] == r[1]["full_steps"] == 2 * 2 * 8
    assert not torch.equal(r[0]["full_actions"], r[1]["full_actions"])
    assert all(np.isfinite(v) for m in r[0]["full_metrics"] for v in m.values())


def test_one_rccl_rank_runs_every_collective_and_changes_no_bit(tmp_path):
    """The multi-GPU path ON RCCL, as far as a 1-GPU box can take it: ONE "nccl" rank with GYMRL_FORCE_COLLECTIVES=1 goes
    through every collective branch of PPO and PPO-full — communicator, parameter broadcast, the reducer's communication
    stream with its two buckets and events, the float64 moments' all-reduce, PPO-full's eager all-reduce between its two
    hipGraphs — and, a sum over one rank being the identity, must end with exactly the bits of a run without any process
    group (worker "none")."""
    N = 64
    outs, res = {}, {}
    for mode in ("nccl", "none"):
        d = tmp_path / mode
        d.mkdir()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("GYMRL_FORCE_COLLECTIVES", None)
        if mode == "nccl":
            env["GYMRL_FORCE_COLLECTIVES"] = "1"
        p = subprocess.run([sys.executable, os.path.join(HERE, "multirank_worker.py"), "0", "1", str(_free_port()), str(d), str(N), mode],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        outs[mode] = p.stdout.decode()
        assert p.returncode == 0, outs[mode]
        res[mode] = torch.load(os.path.join(d, "rank0.pt"), weights_only=False)
    a, b = res["nccl"], res["none"]
    assert a["backend"] == "nccl" and b["backend"] is None
    rs = a["reducer"]                   # 2 epochs x 4 minibatches: two buckets per optimiser step on the communication stream
    assert rs["collectives"] == 2 * 8 and rs["stalls"] == 8 and rs["collective_s"] > 0 and b["reducer"] is None
    for k in ("p0", "params", "moments", "adv", "states", "actions", "log_probs", "values", "rewards", "dones", "full_params",
              "full_actions"):
        assert torch.equal(a[k], b[k]), k
    assert a["metrics"] == b["metrics"] and a["full_metrics"] == b["full_metrics"]
    assert a["full_steps"] == b["full_steps"] == 2 * 2 * 8 and a["full_graphed"] and b["full_graphed"]
