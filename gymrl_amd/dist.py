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


def force_collectives():
    """GYMRL_FORCE_COLLECTIVES=1: run every collective branch with ONE rank as well.  A sum over one rank is the identity,
    so a forced single-rank run must end with the bits of a run without a process group — which is how a 1-GPU box
    executes the whole RCCL path (communicator, the reducer's communication stream, its buckets and events, the moments'
    all-reduce, the parameter broadcast): tests/test_multirank_gpu.py, `bench.py --gpus 1` under that variable."""
    return os.environ.get("GYMRL_FORCE_COLLECTIVES", "0") not in ("", "0")


def collectives_active():
    """True where the trainers issue collectives: more than one rank, or one rank under force_collectives()."""
    return is_dist() and (world_size() > 1 or force_collectives())


def backend():
    return td.get_backend() if is_dist() else None


def init_from_env(backend=None):
    """Initialise from torchrun-style env vars (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world_size, local_rank).  No-op for WORLD_SIZE <= 1 (unless force_collectives())."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lrk = int(os.environ.get("LOCAL_RANK", "0"))
    if (ws > 1 or force_collectives()) and not is_dist():
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
    if collectives_active():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_max(t):
    if collectives_active():
        td.all_reduce(t, op=td.ReduceOp.MAX)
    return t


def broadcast(t, src=0):
    if collectives_active():
        td.broadcast(t, src=src)
    return t


def barrier():
    if collectives_active():
        td.barrier()


def shutdown():
    """Collective teardown: every rank waits for the slowest one, then the process group is destroyed
    (ranks that simply exit while another still runs local work make RCCL's watchdog noisy)."""
    if is_dist():
        if collectives_active():
            td.barrier()
        td.destroy_process_group()


class GradReducer:
    """Bucketed all-reduce(SUM) of a flat gradient buffer on a communication stream.

    The backward pass writes the flat gradient back to front; a bucket [lo, hi) is launched as soon as the kernels that
    write it are queued: the communication stream waits for an event of the compute stream, runs the collective (RCCL puts
    it on its own stream; the communication stream waits for that), and records a completion event.  `wait()` makes the
    compute stream wait for every outstanding bucket — the optimiser calls it right before it reads the gradients, so a
    bucket's collective runs under whatever backward work was queued behind its launch (PPO: the head of the flat buffer
    is reduced under the trunk's three GEMMs).  With `timed=True` every bucket is bracketed by HIP events on the
    communication stream and every wait() by events on the compute stream: `stats()` returns the collectives' own
    durations and the time the compute stream actually stalled on them (bench.py's all-reduce figures)."""

    def __init__(self, flat_grads, bounds=None, timed=False):
        n = flat_grads.numel()
        bounds = [0, n] if bounds is None else sorted({0, n, *[int(b) for b in bounds]})
        if bounds[0] != 0 or bounds[-1] != n:
            raise ValueError("GradReducer: bucket bounds outside the flat buffer")
        self.flat = flat_grads
        self.buckets = [flat_grads[lo:hi] for lo, hi in zip(bounds[:-1], bounds[1:])]
        self.cuda = flat_grads.is_cuda
        self.comm = torch.cuda.Stream(device=flat_grads.device) if self.cuda else None
        self.timed = timed
        self._pending, self._spans, self._stalls = [], [], []

    @property
    def timed(self):
        return self._timed

    @timed.setter
    def timed(self, on):
        """Event timing exists on the device path only: a gloo / CPU reducer stays untimed whatever a trainer assigns
        (stats() would otherwise call torch.cuda.synchronize on a CPU tensor's device)."""
        self._timed = bool(on) and self.cuda

    def launch(self, k):
        """Reduce bucket k; everything queued on the current stream so far is ordered before it."""
        if not collectives_active():
            return
        buf = self.buckets[k]
        if not self.cuda:
            td.all_reduce(buf, op=td.ReduceOp.SUM)
            return
        main = torch.cuda.current_stream(self.flat.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            e0 = None
            if self.timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self.comm)
            td.all_reduce(buf, op=td.ReduceOp.SUM)
            e1 = torch.cuda.Event(enable_timing=self.timed)
            e1.record(self.comm)
        if self.timed:
            self._spans.append((k, e0, e1))
        self._pending.append(e1)

    def wait(self):
        """The current stream waits for every bucket launched since the last wait()."""
        if not self._pending:
            return
        main = torch.cuda.current_stream(self.flat.device)
        s0 = s1 = None
        if self.timed:
            s0 = torch.cuda.Event(enable_timing=True)
            s0.record(main)
        for e in self._pending:
            main.wait_event(e)
        if self.timed:
            s1 = torch.cuda.Event(enable_timing=True)
            s1.record(main)
            self._stalls.append((s0, s1))
        self._pending = []

    def stats(self):
        """{"collectives", "collective_s" (sum of the buckets' own durations), "per_bucket_us", "bucket_bytes",
        "stalls", "stall_s" (compute stream blocked in wait())}; synchronises the device."""
        if not self.timed:
            return None
        torch.cuda.synchronize(self.flat.device)
        per = {}
        for k, e0, e1 in self._spans:
            per.setdefault(k, []).append(e0.elapsed_time(e1) * 1e-3)
        return {"collectives": len(self._spans), "collective_s": sum(sum(v) for v in per.values()),
                "per_bucket_us": {str(k): 1e6 * sum(v) / len(v) for k, v in sorted(per.items())},
                "bucket_bytes": [int(b.numel() * b.element_size()) for b in self.buckets],
                "stalls": len(self._stalls), "stall_s": sum(a.elapsed_time(b) for a, b in self._stalls) * 1e-3}

    def reset_stats(self):
        self._spans, self._stalls = [], []


def rank_devices():
    """Every rank's (rank, local device index, device name, PCI bus id) gathered on all ranks — bench.py prints it so
    that a scaling line shows which devices RCCL actually spanned."""
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        mine = {"rank": rank(), "device": i, "name": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None)}
    else:
        mine = {"rank": rank(), "device": None, "name": "cpu", "pci_bus_id": None}
    if not collectives_active():
        return [mine]
    out = [None] * world_size()
    td.all_gather_object(out, mine)
    return out


def shard_env_ids(num_envs_per_rank, rk=None):
    """Global env-id range owned by a rank (the Philox stream is keyed by env id, so
    an N-GPU run steps exactly the env instances a 1-GPU run with N*num_envs would)."""
    rk = rank() if rk is None else rk
    return rk * num_envs_per_rank, (rk + 1) * num_envs_per_rank
