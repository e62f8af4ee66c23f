"""hipGraph capture of the launch-bound update loops (DQN / Rainbow / SAC at BASELINE configs 3-4).

One off-policy update at batch 128-256 is ~150 launches of a few microseconds each: the GPU idles between
them while Python and the HIP runtime issue the next one.  Captured once as a hipGraph (through
torch.cuda.CUDAGraph, which drives hipStreamBeginCapture on the stream every gymrl_* entry point is handed)
the whole update is ONE graph launch.  What makes a captured update replayable:

  * every tensor it touches has a fixed address: minibatch indices land in a preallocated buffer, the batch
    is gathered from the ring inside the graph, intermediates live in the graph's private pool;
  * the few host scalars that change per step — Adam's bias corrections, the temperature optimiser's — are
    read from a 256-byte device block (`StepScalars`) that the host refreshes with ONE eager launch
    (gymrl_store_scalars: the payload travels as the kernel argument) before each replay;
  * kernels keyed by a host counter (index sampling, ring append, exploration noise) stay eager, in front of
    the replay.

The arithmetic is the eager path's, kernel for kernel, so a graphed run reproduces an eager run bit for bit
(tests/test_graphs_gpu.py).
"""
import contextlib
import gc
import struct

import torch

from . import ops


@contextlib.contextmanager
def capture(graph):
    """`torch.cuda.graph(graph)` with Python's cycle collector held off for the duration of the capture.  A collection
    that runs in the middle of a capture can finalise an OLD trainer (a reference cycle through its closures, kept alive
    until then by a caller's frame); destroying that trainer's captured graphs releases their private memory pool, and a
    device free while a stream is capturing is an error that surfaces inside a destructor — the process aborts.
    (torch's context manager collects once on entry; an additional explicit gc.collect() in front of it was measured to
    make the graph captured afterwards 35 % slower to replay — Rainbow 0.36 -> 0.49 ms per vector step — so there is none.)"""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was_enabled:
            gc.enable()


class StepScalars:
    """Up to 256 bytes of per-step host scalars mirrored into device memory with one launch."""

    SIZE = 256

    def __init__(self, device):
        self.dev = torch.zeros(self.SIZE // 4, dtype=torch.float32, device=device)
        self.host = bytearray(self.SIZE)
        self.used = 0

    def slot(self, nbytes, dtype):
        """Reserve `nbytes` (8-byte aligned) -> (device view of that dtype, byte offset)."""
        off = (self.used + 7) & ~7
        if off + nbytes > self.SIZE:
            raise ValueError("StepScalars block full")
        self.used = off + nbytes
        view = self.dev.view(torch.uint8)[off:off + nbytes].view(dtype)
        return view, off

    def set(self, off, payload):
        self.host[off:off + len(payload)] = payload

    def set_doubles(self, off, *vals):
        self.set(off, struct.pack(f"{len(vals)}d", *vals))

    def flush(self):
        n = (self.used + 3) & ~3
        if n:
            ops.store_scalars(self.dev, bytes(self.host[:n]))


class GraphedStep:
    """fn() run eagerly `warmup` times, then captured once and replayed.  fn must read its per-step inputs
    from fixed device buffers and must not synchronise with the host."""

    def __init__(self, fn, warmup=2):
        self.fn, self.warmup, self.calls, self.graph = fn, warmup, 0, None

    def __call__(self):
        if self.graph is None:
            if self.calls < self.warmup:
                self.calls += 1
                return self.fn()
            self.graph = torch.cuda.CUDAGraph()
            with capture(self.graph):
                self.fn()
        self.graph.replay()


class GraphedUpdate:
    """The pattern the off-policy trainers share: draw the minibatch indices into a fixed buffer (eager, counter-keyed),
    stage every optimiser's bias block (one launch), replay `body(idx, adam_biases, alpha_bias)`.

    optimizers: FusedAdam list (their step counts advance here); alpha = (owner, attr, beta1, beta2) advances the
    integer attribute `attr` of `owner` and stages {1 - beta1^t, 1 - beta2^t} as float64 for a temperature step."""

    def __init__(self, device, batch_size, optimizers, body, alpha=None):
        self.sc = StepScalars(device)
        self.optimizers = list(optimizers)
        slots = [self.sc.slot(16, torch.float32) for _ in self.optimizers]
        self.biases, self.offs = [v for v, _ in slots], [o for _, o in slots]
        self.alpha = alpha
        self.alpha_bias, self.alpha_off = self.sc.slot(16, torch.float64) if alpha else (None, None)
        self.idx = torch.empty(batch_size, dtype=torch.int32, device=device)
        self.step_fn = GraphedStep(lambda: body(self.idx, self.biases, self.alpha_bias))

    def __call__(self, memory, batch_size):
        memory.draw_indices(batch_size, out=self.idx)
        for opt, off in zip(self.optimizers, self.offs):
            self.sc.set(off, opt.next_bias())
        if self.alpha:
            owner, attr, b1, b2 = self.alpha
            t = getattr(owner, attr) + 1
            setattr(owner, attr, t)
            self.sc.set_doubles(self.alpha_off, 1.0 - b1 ** t, 1.0 - b2 ** t)
        self.sc.flush()
        self.step_fn()


class StepChunk:
    """K whole vector steps — acting forward, env step, ring append, index draw, update — as ONE hipGraph.

    A graphed update alone leaves ~25 eager launches per vector step (acting, env, append, draw, noise) whose
    host cost (Python + HIP launch, ~20 us each) is what a step then lasts: 0.50 ms enqueue = 0.50 ms total at
    Rainbow's config 4 (`tools/probe_cpu_bound.py`).  Every one of those launches takes its per-step values —
    ring cursors, Philox counters, PER exponent, Adam's bias corrections — from a record in DEVICE memory instead
    (the `*_dev` arguments of include/gymrl.h), the host walks its own bookkeeping K steps ahead and stages the K
    records with one gymrl_store_scalars, and the K steps replay as one graph: two launches per K vector steps.

    fields: [(name, struct format)] of ONE step's record, e.g. [("noise", "6Q"), ("push", "2q"), ("adam", "4f")];
    every field is 8-byte aligned.  view(j, name) -> the device bytes of that field (a uint8 tensor: pass it as a
    `*_dev` pointer, or .view(dtype) it); set(j, name, *values) writes the host copy; flush() stages all K records.
    """

    LIMIT = 3840

    def __init__(self, device, K, fields):
        self.K, self.fmt, self.off = K, {}, {}
        off = 0
        for name, fmt in fields:
            off = (off + 7) & ~7
            self.fmt[name], self.off[name] = fmt, off
            off += struct.calcsize("=" + fmt)
        self.rec = (off + 7) & ~7
        if K * self.rec > self.LIMIT:
            raise ValueError(f"{K} records of {self.rec} bytes exceed one scalar store ({self.LIMIT} bytes)")
        self.dev = torch.zeros(K * self.rec, dtype=torch.uint8, device=device)
        self.host = bytearray(K * self.rec)
        self.graph = None
        self.key = None

    def view(self, j, name, dtype=None):
        o = j * self.rec + self.off[name]
        v = self.dev[o:o + struct.calcsize("=" + self.fmt[name])]
        return v if dtype is None else v.view(dtype)

    def set(self, j, name, *vals):
        struct.pack_into("=" + self.fmt[name], self.host, j * self.rec + self.off[name], *vals)

    def set_bytes(self, j, name, payload):
        o = j * self.rec + self.off[name]
        self.host[o:o + len(payload)] = payload

    def flush(self):
        ops.store_scalars(self.dev, bytes(self.host))

    def run(self, body, key=None):
        """Replay the captured K-step graph; (re)capture `for j in range(K): body(j)` first when there is none yet or
        `key` (identity of the buffers the body touches) changed.  The records must have been flush()ed."""
        if self.graph is None or key != self.key:
            self.graph, self.key = torch.cuda.CUDAGraph(), key
            with capture(self.graph):
                for j in range(self.K):
                    body(j)
        self.graph.replay()
