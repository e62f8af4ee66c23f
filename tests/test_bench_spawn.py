"""bench.py's multi-rank launch path on CPU: `--gpus N` without torchrun's environment spawns N ranks
itself, the communicator's size must equal N (never silently fewer), and a node with fewer than N devices is refused
with a non-zero exit code before anything is launched.  The workload itself needs MI355Xs; `--spawn-selftest --backend
gloo` runs the launch + verification + one all-reduce without it."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
ENV = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _run(args, env=None, timeout=240):
    return subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, env=env or ENV, cwd=ROOT)


@pytest.mark.timeout(300)
def test_gpus_n_spawns_n_ranks_by_itself():
    r = _run([BENCH, "--gpus", "2", "--backend", "gloo", "--spawn-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["backend"] == "gloo"
    assert out["sum"] == out["expected_sum"] == 3.0                    # both ranks really took part in the collective
    assert sorted(d["rank"] for d in out["rank_devices"]) == [0, 1]
    assert "spawning 2 ranks" in r.stderr


@pytest.mark.timeout(120)
def test_more_gpus_than_devices_is_refused_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    if n == 1:
        n = 2
    r = _run([BENCH, "--gpus", str(n)])
    assert r.returncode == 2
    assert "FATAL" in r.stderr and f"--gpus {n}" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no JSON line pretending n_gpus = n


@pytest.mark.timeout(300)
def test_world_size_mismatch_under_torchrun_is_fatal():
    # the driver's own launch form, but with a --gpus that disagrees with --nproc-per-node
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), BENCH, "--gpus", "3", "--backend", "gloo", "--spawn-selftest"])
    assert r.returncode != 0
    assert "FATAL" in r.stderr and "--gpus 3 but the process group has 2 ranks" in r.stderr


@pytest.mark.timeout(120)
def test_single_rank_selftest_needs_no_process_group():
    r = _run([BENCH, "--gpus", "1", "--backend", "gloo", "--spawn-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["sum"] == 1.0
