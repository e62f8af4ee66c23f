#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of PPO LunarLander-v3 at 4096 envs/GPU
(BASELINE.json configs[1]) + fraction of the HBM roofline on the GAE+loss pass,
with the reference-structured CPU loop timed beside it (cpu_baseline).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full PPO iteration on every rank: a rollout of T vector steps over
N lane-parallel envs (policy forward, categorical sample, LunarLander physics, slab
writes), GAE + advantage moments, and num_epochs x num_minibatches optimiser steps
(minibatch gather, MLP forward/backward, fused clipped-surrogate loss, gradient
all-reduce when N_gpus > 1, clip-norm + Adam).  Nothing is skipped in the timed
region.  value = (T * N * world_size * K) / max-over-ranks wall time.  Weak scaling:
every rank owns its own 4096 envs (global env ids r*N .. (r+1)*N-1).
Data: synthetic — the build's own LunarLander-v3 solver on fixed-seed Philox episodes,
randomly initialised ActorCritic(8, 4, 256).
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC only on this driver: RCCL's P2P set-up between the ranks of a node fails with `hipIpcGetMemHandle: invalid argument`
# under the legacy mode.  Set before the HIP runtime initialises, also when torchrun (not spawn_ranks below) started this process.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, dense f32 MFMA (same guide: 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz)


def full_pass(trainer, T, N, dev, iters=5):
    """SURVEY.md section 8(d)'s definition of the GAE+loss pass: ONE launch of each kernel over the
    whole rollout (T*N transitions, 73 algorithmic bytes each), on the trainer's own slab, HIP
    events around each launch, caches flushed in between (a 1-GiB read) so nothing is served from
    the 256-MB MALL.  Also measures a device-to-device copy for the 'fraction of measured copy
    bandwidth' figure."""
    from gymrl_amd import ops
    buf = trainer.buffer
    B = T * N
    g = torch.Generator(device=dev).manual_seed(2)
    logits = torch.randn(B, 4, device=dev, generator=g)
    v = torch.randn(B, device=dev, generator=g)
    act = buf.actions.view(-1)
    lpo = buf.log_probs.view(-1)
    adv, ret = buf.advantages, buf.returns
    dl, dv = torch.empty_like(logits), torch.empty_like(v)
    met = torch.zeros(5, dtype=torch.float64, device=dev)
    mom = torch.zeros(3, dtype=torch.float64, device=dev)
    nv = trainer._next_value
    cfg = trainer.cfg
    lcfg = trainer._loss_cfg
    flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    src = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    variant = trainer._last_gae_variant      # 2 / 3: the chunk maps (and carries) of the last rollout are still in the workspace

    def timed(fn):
        tot = 0.0
        for _ in range(iters):
            flush.sum()                      # a 1-GiB read: L2 / MALL hold clean lines of something else
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / iters * 1e-3

    gae_s = timed(lambda: ops.gae(buf.rewards, buf.values, buf.dones, nv, cfg.gamma, cfg.gae_lambda,
                                  adv, ret, mom, variant, trainer._gae_ws))
    loss_s = timed(lambda: ops.ppo_loss_fwd_bwd(logits, v, act, lpo, adv.view(-1), ret.view(-1), lcfg, None, mom, dl, dv, met))
    copy_s = timed(lambda: dst.copy_(src))
    copy_bw = 2.0 * src.numel() / copy_s
    bw = 73.0 * B / (gae_s + loss_s)
    return dict(gae_us=round(gae_s * 1e6, 1), gae_variant=variant, loss_us=round(loss_s * 1e6, 1),
                achieved_GBps=round(bw / 1e9, 1), frac=round(bw / HBM_PEAK, 4),
                copy_GBps=round(copy_bw / 1e9, 1), frac_of_copy=round(bw / copy_bw, 4),
                note="one launch per kernel over T*N transitions, cold caches, HIP events per launch")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(a):
    """`python bench.py --gpus N` without torchrun's environment: re-execute this command line as N ranks of ONE node
    through torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and return its exit code.  Refuses,
    loudly and before anything is launched, when the node has fewer than N devices."""
    import subprocess
    if a.backend != "gloo":
        have = torch.cuda.device_count()
        if have < a.gpus:
            print(f"[bench] FATAL: --gpus {a.gpus} needs {a.gpus} visible MI355X devices, this node has {have}; "
                  f"refusing to report an n_gpus={a.gpus} line from fewer devices", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL's P2P set-up needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"[bench] spawning {a.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def verify_world(a, world, local_rank):
    """The process group must be exactly what the command line asked for: `n_gpus` in the JSON line is the size of the
    communicator RCCL built, never a wish.  Raises SystemExit(2) on every rank otherwise."""
    import torch.distributed as td
    got = td.get_world_size() if (td.is_available() and td.is_initialized()) else 1
    problems = []
    if got != a.gpus or world != a.gpus:
        problems.append(f"--gpus {a.gpus} but the process group has {got} ranks (WORLD_SIZE={world})")
    if a.backend != "gloo":
        if got > 1 and td.get_backend() != "nccl":
            problems.append(f"backend is {td.get_backend()}, expected nccl (RCCL)")
        if torch.cuda.device_count() <= local_rank:
            problems.append(f"LOCAL_RANK {local_rank} has no device (device_count = {torch.cuda.device_count()})")
    if problems:
        print("[bench] FATAL: " + "; ".join(problems), file=sys.stderr)
        raise SystemExit(2)
    return got


def spawn_selftest(a, rank, world):
    """--spawn-selftest: the launch + verification path without the workload (CPU-testable with --backend gloo): every
    rank contributes rank + 1 to an all-reduce, rank 0 prints what the ranks agreed on."""
    from gymrl_amd import dist as gdist
    dev = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}") if a.backend != "gloo" else torch.device("cpu")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
    gdist.all_reduce_sum(t)
    devices = gdist.rank_devices()
    if rank == 0:
        import torch.distributed as td
        print(json.dumps({"spawn_selftest": True, "n_gpus": world, "rccl_world_size": world,
                          "backend": td.get_backend() if world > 1 else None, "sum": float(t.item()),
                          "expected_sum": world * (world + 1) / 2.0, "rank_devices": devices}))
        sys.stdout.flush()
    gdist.shutdown()


def _backend_name():
    from gymrl_amd import dist as gdist
    b = gdist.backend() if gdist.collectives_active() else None
    return "nccl (RCCL)" if b == "nccl" else b


def comm_summary(world, reducer, ks, step_s, steps):
    """All-reduce figures of the timed region (SURVEY.md 8(d) "Grad all-reduce" row): the collectives' own durations on the
    communication stream, and how long the compute stream actually stalled on them."""
    from gymrl_amd import dist as gdist
    if not gdist.collectives_active():
        return {"rccl_world_size": 1, "grad_allreduce": None, "moments_allreduce": None}
    out = {"rccl_world_size": world}
    if world == 1:
        out["forced"] = ("GYMRL_FORCE_COLLECTIVES=1: ONE rank runs every collective of the multi-GPU path through RCCL (sums over "
                         "one rank: the result's bits are those of a run without a process group); the figures are the path's "
                         "fixed cost per collective on this box, not an xGMI measurement")
    st = reducer.stats() if reducer is not None else None
    if st:
        n = max(st["collectives"], 1)
        out["grad_allreduce"] = {
            "collectives_per_step": st["collectives"] / steps, "bucket_bytes": st["bucket_bytes"],
            "avg_us_per_collective": round(1e6 * st["collective_s"] / n, 2), "per_bucket_us": {k: round(v, 2) for k, v in st["per_bucket_us"].items()},
            "us_per_step": round(1e6 * st["collective_s"] / steps, 1), "pct_of_step": round(100.0 * st["collective_s"] / steps / step_s, 3),
            "exposed_us_per_step": round(1e6 * st["stall_s"] / steps, 1), "exposed_pct_of_step": round(100.0 * st["stall_s"] / steps / step_s, 3),
            "note": "HIP events on the communication stream around every bucket (own duration, incl. the wait for the compute "
                    "stream's gradient kernels to drain out of the device) / on the compute stream around GradReducer.wait() (exposed)"}
    if ks and "moments_allreduce" in ks:
        m = ks["moments_allreduce"]
        out["moments_allreduce"] = {"us_per_call": round(m["avg_us"], 2), "calls_per_step": m["launches"] / steps, "bytes": 24}
    return out


def _event_us(fn, reps=20, warm=3):
    """Average duration of fn() in microseconds, HIP events on torch's current stream (the C-ABI's launch stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def _random_access_GBps(dev, entries=1 << 21, draws=1 << 22):
    """What this box gives 8-byte reads at random addresses of a 16-MiB float64 array (the sum tree's size at 2^20 leaves):
    the yardstick SURVEY.md 8(d) asks PER sample / update to be priced against.  Useful bytes only (8 per access)."""
    g = torch.Generator(device=dev).manual_seed(3)
    table = torch.rand(entries, dtype=torch.float64, device=dev, generator=g)
    idx = torch.randint(0, entries, (draws,), device=dev, generator=g)
    out = torch.empty(draws, dtype=torch.float64, device=dev)
    us = _event_us(lambda: torch.index_select(table, 0, idx, out=out), reps=10)
    return draws * 8.0 / us / 1e3, draws / us * 1e6


def main_offpolicy(a, rank, world, local_rank):
    """BASELINE.json configs[2] (Rainbow DQN CartPole-v1, 8192 envs, PER sum tree + n-step + NoisyNet) and configs[3] (SAC
    Pendulum-v1, 4096 envs, twin Q + reparameterised sample + automatic temperature), one GPU each; with --gpus N every rank
    runs an independent replica (the off-policy path has no exchange step: DESIGN.md section 6, "replicas only").
    One "step" = 16 vector steps = one replay of the trainer's StepChunk graph where the trainer chunks (Rainbow), 16 passes
    of its loop otherwise: per vector step the reference loop's acting forward, env.step, n-step / ring store, proportional or
    uniform draw and ONE update (rainbow_dqn_cartpole.py:363-405, sac_pendulum.py:275-300), ring of 2^20 rows.
    Nothing is skipped.  After the timed region rank 0 event-times the replay memory's pieces at the run's own sizes and at
    a throughput size, against this box's measured random-access rate."""
    from gymrl_amd import dist as gdist
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    VS = 16
    if a.algo == "rainbow":
        from gymrl_amd.rainbow_dqn_cartpole import Config, RainbowDQNTrainer as Trainer
        N, B, env_name = a.envs, a.batch or 256, "CartPole-v1"
    else:
        from gymrl_amd.sac_pendulum import Config, SACTrainer as Trainer
        N, B, env_name = a.envs, a.batch or 128, "Pendulum-v1"
    cfg = Config()
    cfg.num_envs, cfg.memory_capacity, cfg.max_episodes, cfg.batch_size, cfg.seed = N, 1 << 20, 10 ** 9, B, 0
    cfg.device = str(dev)
    sys.stdout = open(os.devnull, "w") if rank != 0 else sys.stderr
    tr = Trainer(cfg)
    from gymrl_amd.envs import VecEnv

    # ONE env object for the warm-up and the timed call (VecEnv.close() only joins its side stream, the object stays usable): the
    # StepChunk graph is keyed by the env's identity and state address, and a fresh env per call made the timed region start
    # with a re-capture whenever the allocator did not hand the old address back (0.099 vs 0.123 ms per SAC vector step, same kernels)
    env = VecEnv(env_name, N, device=dev, seed=tr.base_seed, env_id0=rank * N)

    def run(vector_steps):
        tr.env = env
        tr.episode_rewards.clear()            # (Rainbow's loop stops at a solved running mean; the bench times a fixed count)
        tr.train(max_vector_steps=vector_steps)
    run(max(a.warmup, 1) * VS + 64)           # fills the n-step windows and the ring past one batch, captures the graphs
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps * VS)
    gdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    gdist.all_reduce_max(dt_t)
    dt = float(dt_t.item())
    devices = gdist.rank_devices()
    if rank != 0:
        gdist.shutdown()
        return
    sys.stdout = sys.__stdout__
    try:
        ra_GBps, ra_per_s = _random_access_GBps(dev)
        mem = tr.memory
        cap = 1 << 20
        depth = 21                                                  # node reads of one descent through a 2^20-leaf tree
        pieces = {}
        if a.algo == "rainbow":
            from gymrl_amd import ops
            D = 4
            row_bytes = 4 * (2 * D + 2) + 1 + 4                     # state, next_state, action, reward, flag + the index
            for nb in sorted({B, 8192, 65536}):
                idx = torch.empty(nb, dtype=torch.int32, device=dev)
                pr, w = torch.empty(nb, dtype=torch.float64, device=dev), torch.empty(nb, device=dev)
                ws = ops.per_workspace(max(8192, cap, nb), dev)
                k = [0]

                def draw():
                    k[0] += 1
                    ops.per_sample(mem.sum_tree.tree, cap, nb, mem.current_size, 0.5, ws, seed=7, counter=k[0], out=(idx, pr, w))
                us = _event_us(draw)
                pieces[f"per_sample B={nb}"] = dict(us=round(us, 2), draws_per_s=round(nb / us * 1e6), bytes_per_draw=8 * depth,
                                                    GBps=round(8.0 * depth * nb / us / 1e3, 2),
                                                    frac_of_random_access=round(8.0 * depth * nb / us / 1e3 / ra_GBps, 4))
                us = _event_us(lambda: ops.replay_gather(mem.ring, idx))
                pieces[f"ring_gather B={nb}"] = dict(us=round(us, 2), rows_per_s=round(nb / us * 1e6), bytes_per_row=row_bytes,
                                                     GBps=round(row_bytes * nb / us / 1e3, 2), frac_hbm=round(row_bytes * nb / us / 1e3 / (HBM_PEAK / 1e9), 5))
                td = torch.randn(nb, device=dev)
                ws2 = ops.per_workspace(max(8192, cap, nb), dev)

                def upd():
                    ops.per_update(mem.sum_tree.tree, cap, nb, ws2, idx=idx, prio=ops.per_priorities(td, mem.alpha, 0.01))
                if nb <= 8192:
                    us = _event_us(upd)
                    pieces[f"per_update B={nb} (priorities of a sampled batch, reference order per node)"] = dict(
                        us=round(us, 2), indices_per_s=round(nb / us * 1e6), bytes_per_index=16 * (depth - 1),
                        GBps=round(16.0 * (depth - 1) * nb / us / 1e3, 3))
            us = _event_us(lambda: mem.sum_tree.update_range(0, N, priority=1.0))
            store_bytes = 16.0 * (depth - 1) * N
            pieces[f"per_update N={N} (the new rows of one vector step)"] = dict(
                us=round(us, 2), indices_per_s=round(N / us * 1e6), bytes_per_index=16 * (depth - 1), GBps=round(store_bytes / us / 1e3, 3))
            roof = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK / 1e9, achieved=round(store_bytes / us / 1e3, 3),
                        frac=round(store_bytes / us / 1e3 / (HBM_PEAK / 1e9), 6), traffic=None, launch_s=us * 1e-6,
                        kernel="gymrl_per_update over the N new rows of a vector step (csrc/per.hip per_store_leaf + per_store_ancestor): "
                               "320 algorithmic bytes per index (20 float64 read-modify-writes).  Round 3 applied a node's additions in "
                               "batch order (the root: one dependent chain of N float64 adds, 72-79 us); the reference stores ONE row per "
                               "step, so the N-row store is now defined as one pairwise-summed addition per ancestor (DESIGN.md section 4, "
                               "identical to the reference at N = 1).  Latency-bound scattered 8-byte accesses: priced against HBM as the "
                               "contract asks, and against this box's random-access rate below",
                        random_access_GBps_measured=round(ra_GBps, 1), random_accesses_per_s_measured=round(ra_per_s),
                        frac_of_random_access=round(store_bytes / us / 1e3 / ra_GBps, 5))
            # the fused step's own launches (csrc/offpolicy_step.hip), event-timed at the run's sizes
            if tr._fused_act_ok():
                lb = tr._loop_buffers(N, 4)
                trk = lb["tracker"]
                us = _event_us(lambda: tr._vector_step(lb, lb["obs"], lb["nxt"], trk.ret[0], trk.done[0]), reps=20)
                H = cfg.hidden_dim
                fl = 2.0 * N * (4 * H + H * H + 3 * H)
                pieces["acting step: noisy heads + fc1, fc2, dueling argmax + CartPole + n-step push (gymrl_noisy_combine + gymrl_rainbow_act_step) "
                       "+ the new rows' sum-tree store on the side stream"] = dict(us=round(us, 2), TFLOPs=round(fl / us / 1e6, 2))
                upd_us = _event_us(lambda: tr.update_async(), reps=30)
                flu = 2.0 * B * (3 * (4 * H + H * H + 3 * H) + (3 * H + H * H) + (3 * H + H * H + 4 * H))
                cus = min(256, 3 * ((B + 15) // 16))
                pieces[f"one update at batch {B} (proportional draw, three forwards, TD loss, backward, clip + Adam + Polyak, update_priorities)"] = dict(
                    us=round(upd_us, 2), TFLOPs=round(flu / upd_us / 1e6, 3), frac_mfma=round(flu / upd_us / 1e6 / (MFMA_F32_PEAK / 1e12), 5),
                    note=f"gymrl_rainbow_update's row kernel carries 16 rows per workgroup, three workgroups per slab: {cus} of the 256 compute units do its MFMA work")
                if B > 256:      # SURVEY 8(d)'s throughput-sized line: the update is the dominant launch group, priced against the f32-MFMA peak
                    pieces[f"per_update N={N} (the new rows of one vector step): the small-batch line's roofline"] = roof
                    roof = dict(bound="mfma", unit="TFLOP/s", peak=MFMA_F32_PEAK / 1e12, achieved=round(flu / upd_us / 1e6, 3),
                                frac=round(flu / upd_us / 1e6 / (MFMA_F32_PEAK / 1e12), 5), traffic=None, launch_s=upd_us * 1e-6,
                                flops_per_launch_group=flu, compute_units_carrying_rows=cus,
                                kernel=f"one Rainbow update at batch {B}: gymrl_rainbow_update's row-slab kernel ({3 * ((B + 15) // 16)} workgroups of 16 rows on slab-adjacent "
                                       "1-D grids) + the weight-gradient tiles in lin.hip's 256-row slices + clip / Adam / Polyak + update_priorities "
                                       "(sort-based passes above 512 indices)")
        else:
            from gymrl_amd import ops
            D, A = 3, 1
            row_bytes = 4 * (2 * D + A + 1) + 1 + 4
            for nb in (B, 65536):
                idx = ops.uniform_indices(7, 1, len(mem), min(nb, len(mem)), dev)
                nb = idx.numel()
                us = _event_us(lambda: ops.replay_gather(mem.ring, idx))
                pieces[f"ring_gather B={nb}"] = dict(us=round(us, 2), rows_per_s=round(nb / us * 1e6), bytes_per_row=row_bytes,
                                                     GBps=round(row_bytes * nb / us / 1e3, 2), frac_hbm=round(row_bytes * nb / us / 1e3 / (HBM_PEAK / 1e9), 5))
            H = cfg.hidden_dim
            DA = D + A
            # algorithmic f32 flops of one update (sac_pendulum.py:213-267): actor(s'), target Q x2, Q x2 forward + input gradients,
            # actor(s), Q x2 forward + input gradients back to the action and through the actor, every weight gradient
            actor_f, q_f = 2.0 * B * (D * H + H * H + 2 * H * A), 2.0 * B * (DA * H + H * H + H)
            flu = (actor_f + 2 * q_f + 2 * q_f + 2 * 2.0 * B * (H + H * H) + 2 * q_f            # P1 + critic dW
                   + actor_f + 2 * q_f + 2 * 2.0 * B * (H + H * H + DA * H) + 2.0 * B * (2 * A * H + H * H) + actor_f)   # P3 + actor dW
            upd_us = _event_us(lambda: tr.update_async() if getattr(cfg, "use_graphs", True) else tr.update(), reps=50)
            cus = min(256, 4 * ((B + 15) // 16))      # P1: four workgroups per 16-row slab (P3: two)
            roof = dict(bound="mfma", unit="TFLOP/s", peak=MFMA_F32_PEAK / 1e12, achieved=round(flu / upd_us / 1e6, 3),
                        frac=round(flu / upd_us / 1e6 / (MFMA_F32_PEAK / 1e12), 5), traffic=None, launch_s=upd_us * 1e-6,
                        flops_per_launch_group=flu, compute_units_carrying_rows=cus,
                        frac_of_those_units=round(flu / upd_us / 1e6 / (MFMA_F32_PEAK / 1e12 * cus / 256.0), 4),
                        kernel=f"gymrl_sac_update: the four launches of one update at batch {B} (csrc/offpolicy_step.hip: row-slab kernels P1 / P3, "
                               "weight-gradient + Adam tile kernels P2 / P4).  A 16-row slab's chain of layers runs on ONE compute unit, its independent "
                               "chains on units of their own: 4 B / 16 (P1: 32 at batch 128) and 2 B / 16 (P3) workgroups carry the MFMA "
                               "work of the row kernels (a 16 x 256 x 256 layer is 3.9 us of f32 MFMA on one compute unit, 4.6-4.9 us measured: "
                               "tools/probe_sac_stages.py), so the figure is priced twice — against the chip's f32-MFMA peak as the contract asks "
                               "(`frac`) and against the peak of the units that can work (`frac_of_those_units`: P1's workgroups, at most 256)")
            pieces[f"one update at batch {B} (twin critics, actor, temperature, Adam x3 + Polyak: gymrl_sac_update, 4 launches)"] = dict(us=round(upd_us, 2))
            if tr._fused_ok():
                lb = tr._loop_buffers(N, D)
                trk = lb["tracker"]
                us = _event_us(lambda: tr._vector_step(lb, lb["obs"], lb["nxt"], trk.ret[0], trk.done[0]), reps=20)
                fl = 2.0 * N * (D * H + H * H + 2 * H * A)
                pieces["acting step: Actor forward + reparameterised draw + Pendulum step + replay row (gymrl_sac_act_step, 1 launch)"] = dict(
                    us=round(us, 2), TFLOPs=round(fl / us / 1e6, 2), frac_mfma=round(fl / us / 1e6 / (MFMA_F32_PEAK / 1e12), 4))
        vsteps = a.steps * VS
        out = {
            "metric": f"env-steps/sec at N envs/GPU ({'Rainbow DQN CartPole' if a.algo == 'rainbow' else 'SAC Pendulum'}), 1 update per vector step + %roofline",
            "value": N * vsteps * world / dt, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (sum tree f64)" if a.algo == "rainbow" else "f32 (temperature f64)", "data": "synthetic",
            "config": {"workload": (f"Rainbow DQN CartPole-v1, {N} envs, PER sum tree (2^20 leaves) + 5-step returns + NoisyNet, batch {B} "
                                    + ("(BASELINE.json configs[2])" if not a.batch else "(configs[2] at SURVEY 8(d)'s throughput-sized batch)") if a.algo == "rainbow" else
                                    f"SAC Pendulum-v1, {N} envs, twin Q + reparameterised sample + automatic temperature, ring 2^20, batch {B} "
                                    + ("(BASELINE.json configs[3])" if not a.batch else "(configs[3] at SURVEY 8(d)'s throughput-sized batch)")),
                       "envs_per_gpu": N, "vector_steps_per_step": VS, "updates_per_vector_step": 1, "batch": B, "replay_rows": cap,
                       "ms_per_vector_step": round(dt / vsteps * 1e3, 4), "updates_per_s": round(vsteps * world / dt, 1),
                       "parallelism": f"{world} independent replicas (no exchange step on this path)" if world > 1 else "single GPU"},
            "roofline": roof, "pieces": pieces,
            "comm": dict(rccl_world_size=world, grad_allreduce=None, backend=_backend_name(), rank_devices=devices),
        }
        print(json.dumps(out))
        sys.stdout.flush()
    finally:
        gdist.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 3; 100 for --algo sac / rainbow, whose step — 16 vector steps, ~2.3 ms — is short "
                         "enough for the host's first chunks to move the figure by 15 %% below ~50)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 1; 20 for --algo sac / rainbow)")
    ap.add_argument("--envs", type=int, default=None,
                    help="env instances per GPU (default: the BASELINE config's — 4096 for ppo / ppo_full / sac, 8192 for rainbow)")
    ap.add_argument("--batch", type=int, default=None,
                    help="sac / rainbow: rows per update (default: the reference's 128 / 256; SURVEY 8(d)'s throughput-sized lines: 4096 / 8192)")
    ap.add_argument("--rollout", type=int, default=2048, help="T: vector steps per rollout (reference update_freq)")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--minibatches", type=int, default=32)
    ap.add_argument("--algo", choices=("ppo", "ppo_full", "rainbow", "sac"), default="ppo",
                    help="ppo = the headline benchmark (BASELINE configs[1]); ppo_full = configs[4]'s per-GPU workload; "
                         "rainbow / sac = configs[2] / configs[3] (a step = 16 vector steps with one update each)")
    ap.add_argument("--micro-batch", type=int, default=524288,
                    help="ppo_full: rows per forward/backward pass (524288: each launch's fixed cost — weight staging, partial sums — halves against 262144; larger is flat)")
    ap.add_argument("--timer-every", type=int, default=8,
                    help="ppo: HIP-event brackets around the launches of every K-th minibatch of the timed region (1: all; "
                         "the gather of the next minibatch is issued inside the previous one's bracket window either way)")
    ap.add_argument("--graphs", action="store_true", help="ppo: replay the minibatch body as a hipGraph (A/B; measured 1 % slower than the eager queue)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=24.0)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="nccl == RCCL (default); gloo only for --spawn-selftest on CPU")
    ap.add_argument("--spawn-selftest", action="store_true", help="launch + verify the ranks, all-reduce once, print one JSON line; no workload")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.envs is None:
        a.envs = 8192 if a.algo == "rainbow" else 4096
    short = a.algo in ("sac", "rainbow")
    if a.steps is None:
        a.steps = 200 if short else 3
    if a.warmup is None:
        # sac / rainbow: a step is 16 vector steps = ~1.5 ms of a FEW compute units' work, and the shader clock climbs from its
        # idle level over the first ~0.2 s of such a load (measured: 20 warm-up steps = 30 ms gave 33-34 M env-steps/s for SAC in
        # every process but a box's first, 41.7 M there and 39 M over 400 timed steps) — the default warm-up covers the ramp
        a.warmup = 300 if short else 1
    if a.envs < 1:
        ap.error("--envs must be >= 1")
    if a.backend == "gloo" and not a.spawn_selftest:
        ap.error("--backend gloo is only for --spawn-selftest (the benchmark itself runs on RCCL)")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))            # N ranks re-enter main() with torchrun's environment

    from gymrl_amd import dist as gdist
    rank, world, local_rank = gdist.init_from_env(backend=a.backend if a.backend == "gloo" else None)
    verify_world(a, world, local_rank)
    if a.spawn_selftest:
        return spawn_selftest(a, rank, world)
    if a.algo == "ppo_full":
        return main_ppo_full(a, rank, world, local_rank)
    if a.algo in ("rainbow", "sac"):
        return main_offpolicy(a, rank, world, local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from gymrl_amd.ppo_lunarlander import Config, KernelTimers, PPOTrainer
    cfg = Config()
    cfg.env_name = "LunarLander-v3"
    cfg.seed = 0                       # fixed-seed episodes
    cfg.num_envs = a.envs
    cfg.update_freq = a.rollout
    cfg.num_epochs = a.epochs
    cfg.num_minibatches = a.minibatches
    cfg.max_train_steps = 10**12       # keep the LR anneal well-defined for any K
    cfg.device = str(dev)
    cfg.use_graphs = bool(a.graphs)
    sys.stdout = open(os.devnull, "w") if rank != 0 else sys.stderr       # stdout carries the JSON line only
    trainer = PPOTrainer(cfg)
    T, N = cfg.update_freq, cfg.num_envs
    init_params = trainer.flat_params.clone()

    phase_events = []

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        phase_events.append(ev)
        ev[0].record()
        if cfg.anneal_lr:
            lr = cfg.lr * (1.0 - trainer.step_count * world / cfg.max_train_steps)
            for g in trainer.optimizer.param_groups:
                g["lr"] = lr
        nv = trainer.collect_rollout()
        ev[1].record()
        m = trainer.update(nv)
        ev[2].record()
        return m

    for _ in range(a.warmup):
        step()
    phase_events.clear()
    timers = KernelTimers(every=max(1, a.timer_every))
    trainer._timers = timers
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    metrics = None
    for _ in range(a.steps):
        metrics = step()
    gdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    gdist.all_reduce_max(dt_t)
    dt = float(dt_t.item())
    trainer._timers = None
    devices = gdist.rank_devices()     # collective: every rank's device, as the communicator sees them

    if rank != 0:
        gdist.shutdown()               # waits for rank 0's post-processing, then leaves together
        return
    sys.stdout = sys.__stdout__
    try:
        # rollout under the initial policy (outside the timed region; the trained parameters are put back)
        trained = trainer.flat_params.clone()
        trainer.flat_params.copy_(init_params)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sc, rc = trainer.step_count, trainer.rollout_count
        e0.record()
        trainer.collect_rollout()
        e1.record()
        torch.cuda.synchronize()
        frozen_ms = round(e0.elapsed_time(e1), 1)
        trainer.flat_params.copy_(trained)
        trainer.step_count, trainer.rollout_count = sc, rc
        report(a, cfg, trainer, timers, phase_events, metrics, dt, world, T, N, dev, frozen_ms, devices)
    finally:
        gdist.shutdown()               # always release the other ranks, even if the report fails


def _sub_bwd_launch(rows, D, sk_it, dev, reps=5):
    """Average duration (HIP events on the launch stream) of gymrl_mhc_sub_backward at `rows` rows of a 2 x D branch stack, and
    its algorithmic bytes per row: g, h, d_h [2, D] + z, d_z [D] + the row's saved gates and sums (2 + 2 + 4 + 9 floats)."""
    from gymrl_amd import ops
    n = 2
    h, g = torch.randn(rows, n, D, device=dev), torch.randn(rows, n, D, device=dev)
    nw, w = torch.rand(n * D, device=dev) + 0.5, torch.randn(n * D, 8, device=dev) * 0.3
    alpha, beta = torch.tensor([0.7, -0.4, 0.9], device=dev), torch.randn(8, device=dev) * 0.1
    W, b = torch.randn(D, D, device=dev) / D ** 0.5, torch.zeros(D, device=dev)
    pre, post, mix, stats, read, z, _ = ops.mhc_sub_forward(h, nw, w, alpha, beta, W, b, sk_it)
    run = lambda: ops.mhc_sub_backward(g, h, z, pre, post, mix, stats, nw, w, alpha, W)  # noqa: E731
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    per_row = 4 * (3 * n * D + 2 * D + 17)
    return {"s": e0.elapsed_time(e1) * 1e-3 / reps, "rows": rows, "bytes_per_row": per_row, "bytes": per_row * rows}


def main_ppo_full(a, rank, world, local_rank):
    """BASELINE.json configs[4]: PPO-full (mHC network, decoupled-lambda GAE, entropy-ratio mask) on LunarLander-v3 at
    4096 envs per GPU with the flat-gradient all-reduce.  One step = one iteration of ppo_full_lunarlander.py:681-700 on
    every rank: collect_experience (T vector steps: mHC forward replayed as a hipGraph, categorical draw + behaviour
    entropy + both GAE chunk maps, LunarLander step), G3 (carry + apply), F0's 4 epochs x 4 minibatches (the reference's
    4096 / 1024 ratio: T*N/4 rows each, accumulated over micro-batches of --micro-batch rows) with the L3 loss kernel,
    clip + Adam.  The network's rollout forward is one hand-written launch per vector step; its training pass is autograd over the
    C-ABI's HBM-bound kernels (csrc/mhc.hip, csrc/lin.hip) with the library's GEMMs for the 128-wide forward / input gradients.
    `roofline` = the largest of those kernels, the gates backward, event-timed at the micro-batch size after the timed region."""
    from gymrl_amd import dist as gdist
    from gymrl_amd.ppo_full_lunarlander import Config, PPOTrainer
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    cfg = Config()
    cfg.seed, cfg.num_envs, cfg.device = 0, a.envs, str(dev)
    cfg.update_freq = a.rollout if a.rollout != 2048 else 4096          # F0: update_freq 4096 per env
    cfg.num_epochs, cfg.num_minibatches = (a.epochs if a.epochs != 10 else 4), (a.minibatches if a.minibatches != 32 else 4)
    cfg.micro_batch = a.micro_batch
    cfg.max_train_steps = 10**12
    sys.stdout = open(os.devnull, "w") if rank != 0 else sys.stderr       # stdout carries the JSON line only
    torch.manual_seed(0)
    tr = PPOTrainer(cfg)
    T, N = cfg.update_freq, cfg.num_envs
    ev_all = []

    def step():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev_all.append(ev)
        ev[0].record()
        tr.collect_experience()
        ev[1].record()
        adv, ret = tr.compute_advantages()
        ev[2].record()
        m = tr.update_model(adv, ret)
        ev[3].record()
        return m
    if True:
        for _ in range(a.warmup):
            step()
        ev_all.clear()
        tr._time_collectives = True
        if tr._reducer is not None:
            tr._reducer.reset_stats()
        gdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = None
        for _ in range(a.steps):
            m = step()
        gdist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device=dev)
    gdist.all_reduce_max(dt_t)
    dt = float(dt_t.item())
    devices = gdist.rank_devices()
    if rank == 0:
        sys.stdout = sys.__stdout__
        ph = [sum(e[i].elapsed_time(e[i + 1]) for e in ev_all) / a.steps for i in range(3)]
        gae_s = ph[1] * 1e-3
        mb = T * N // cfg.num_minibatches
        gates = _sub_bwd_launch(min(cfg.micro_batch, mb), cfg.mhc_dim, cfg.mhc_sk_it, dev) if (cfg.mhc_rate == 2 and cfg.mhc_dim == 128) else None
        out = {
            "metric": "env-steps/sec at N envs/GPU (PPO-full LunarLander), 1/2/4/8 GPUs + %HBM roofline",
            "value": T * N * world * a.steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PPO-full LunarLander-v3, 4096 envs per MI355X (BASELINE.json configs[4], per-GPU shard)",
                       "envs_per_gpu": N, "rollout_len": T, "num_epochs": cfg.num_epochs, "num_minibatches": cfg.num_minibatches,
                       "minibatch": mb, "micro_batch": min(cfg.micro_batch, mb),
                       "optimizer_steps_per_iteration": cfg.num_epochs * cfg.num_minibatches,
                       "parallelism": f"dp{world} (env shards + flat-gradient all-reduce)" if world > 1 else "single GPU"},
            "roofline": ({"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
                          "kernel": "gymrl_mhc_sub_backward (csrc/mhc.hip: combine, the Linear's input gradient on f32 MFMA, read and gates "
                                    "backward of one hyper-connection sub-block in one launch): the largest kernel of the training "
                                    "pass, four launches per micro-batch; bound by the CU's memory pipe and 352 f32 MFMAs per 16 rows",
                          "achieved": round(gates["bytes"] / gates["s"] / 1e9, 1), "frac": round(gates["bytes"] / gates["s"] / HBM_PEAK, 4),
                          "traffic": None, "launch_s": gates["s"], "rows": gates["rows"], "bytes_per_row": gates["bytes_per_row"],
                          "note": "event-timed after the timed region at the micro-batch size (cold inputs of that size: 0.9 GB)"}
                         if gates else None),
            "gae_pass": {"kernel": "G3: gymrl_gae_decoupled variant 2 (two carry scans + apply; both chunk maps composed in the rollout)",
                         "GBps": round(17.0 * T * N / gae_s / 1e9, 1), "frac": round(17.0 * T * N / gae_s / HBM_PEAK, 4), "launch_s": gae_s},
            "phases": {"rollout_ms": round(ph[0], 1), "gae_ms": round(ph[1], 3), "update_ms": round(ph[2], 1)},
            "train_metrics": {k: float(v) for k, v in (m or {}).items()},
            "comm": dict(comm_summary(world, tr._reducer, None, dt / a.steps, a.steps), backend=_backend_name(),
                         rank_devices=devices),
        }
        print(json.dumps(out))
        sys.stdout.flush()
    gdist.shutdown()


def report(a, cfg, trainer, timers, phase_events, metrics, dt, world, T, N, dev, frozen_ms=None, devices=None):
    ks = timers.summary()
    transitions = T * N
    rollout_s = sum(e[0].elapsed_time(e[1]) for e in phase_events) * 1e-3 / a.steps
    update_s = sum(e[1].elapsed_time(e[2]) for e in phase_events) * 1e-3 / a.steps
    Cw, Dw, Aw = cfg.hidden_dim, 8, 4
    fu = getattr(trainer, "_fused_update", None)
    # algorithmic bytes per unit of the HBM-bound kernels: GAE 17 B/elt, loss
    # fwd+bwd 56 B/sample, gather 64 B line in + 48 B out + 4 B index, Adam 36 B/param incl. norm pre-pass + zero_grad;
    # per row of a [B, C = 256] activation: linear_tanh_smallk read 4D, write 4C; heads_loss_fwd_bwd read 8C + 16 (act,
    # logp_old, adv, ret), write 8C; linear_smallk_bwd (dZ given) read 4C + 4D; round-1 passes as in round 1
    bytes_per_unit = {"gae": 17.0, "ppo_loss_fwd_bwd": 56.0, "gather_minibatch": 116.0, "adam_step": 36.0,
                      "linear_tanh_smallk": 4.0 * (Dw + Cw), "tanh_inplace": 8.0,
                      "heads_loss_fwd_bwd": 16.0 * Cw + 16.0,
                      "heads_fwd_tanh": 8.0 * Cw + 4.0 * (Aw + 1), "heads_bwd": 16.0 * Cw + 4.0 * (Aw + 1),
                      "tanh_bwd_colsum": 12.0 * Cw,
                      "linear_smallk_bwd": (4.0 if "gemm_dx_256_tanhbwd" in ks else 8.0) * Cw + 4.0 * Dw}
    # flops per row of the hand-written f32-MFMA GEMMs (csrc/gemm.hip)
    flops_per_unit = {"gemm_fwd_256_tanh": 2.0 * Cw * Cw, "gemm_fwd_512": 4.0 * Cw * Cw, "gemm_dw_512": 4.0 * Cw * Cw,
                      "gemm_dx_512_tanhbwd": 4.0 * Cw * Cw, "gemm_dw_256_db": 2.0 * Cw * Cw, "gemm_dx_256_tanhbwd": 2.0 * Cw * Cw}
    kernels = {}
    for k, v in ks.items():
        ent = dict(launches=v["launches"], avg_us=round(v["avg_us"], 2))
        if k in bytes_per_unit:
            gbps = bytes_per_unit[k] * v["units"] / v["total_s"] / 1e9
            ent.update(bound="hbm", bytes_per_unit=bytes_per_unit[k], achieved_GBps=round(gbps, 1), frac=round(gbps * 1e9 / HBM_PEAK, 4))
        if k in flops_per_unit:
            tf = flops_per_unit[k] * v["units"] / v["total_s"] / 1e12
            ent.update(bound="mfma", flops_per_unit=flops_per_unit[k], achieved_TFLOPs=round(tf, 1), frac=round(tf * 1e12 / MFMA_F32_PEAK, 4))
        kernels[k] = ent
    # counter-measured HBM traffic lives in a separate rocprofv3 --pmc pass (the tool cannot run inside this process); its
    # summary is attached only when it was taken on THIS binary: the sha256 of libgymrl_hip.so stamped into the summary
    # (tools/pmc_gemm.py -> tools/pmc_gemm_summarise.py) must equal the library loaded here, else the summary is refused
    from gymrl_amd import _lib
    lib_hash = _lib.lib_sha256()
    prof, prof_refused = None, None
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json")), reverse=True):
        summary = json.load(open(os.path.join(ROOT, "profiles", name)))
        stamp = (summary.get("provenance") or {}).get("libgymrl_hip_sha256")
        if stamp == lib_hash:
            prof = (name, summary)
            break
        if prof_refused is None:
            prof_refused = dict(file="profiles/" + name, refused=("no provenance stamp" if stamp is None else
                                                                  f"taken on library {stamp[:16]}..., this run loaded {lib_hash[:16]}..."))
    gemms = [k for k in flops_per_unit if k in ks]
    n_upd = a.steps * cfg.num_epochs * cfg.num_minibatches
    if not gemms:
        raise SystemExit("[bench] the headline workload runs ppo_net.FusedActorCriticUpdate.step() (hidden_dim 256); no GEMM timers found")
    # The dominant hand-written work of the step: the six exact-f32 MFMA GEMM launches of every minibatch update
    # (csrc/gemm.hip), event-timed on the launch stream inside the timed region.
    # (per launch: the timers bracket every --timer-every-th minibatch)
    fl = sum(flops_per_unit[k] * ks[k]["units"] / ks[k]["launches"] for k in gemms)
    sec = sum(ks[k]["total_s"] / ks[k]["launches"] for k in gemms)
    head = dict(bound="mfma", unit="TFLOP/s", peak=MFMA_F32_PEAK / 1e12, achieved=round(fl / sec / 1e12, 1),
                frac=round(fl / sec / MFMA_F32_PEAK, 4),
                kernel="the six f32-MFMA GEMM launches of one minibatch update (csrc/gemm.hip: gemm_ws_kernel x4, gemm_tn_kernel x2 "
                       "+ their reductions), algorithmic flops / summed average launch durations",
                flops_per_launch_group=fl, launch_s=sec, share_of_step=round(sec * n_upd / a.steps / (rollout_s + update_s), 3))
    # The pass BASELINE.json's metric names (GAE + clipped-surrogate loss).  In the run the loss is no longer a pass of its
    # own: it is evaluated inside gymrl_heads_loss_fwd_bwd on values that never leave registers, so `in_run` reports the GAE
    # launch (17 B per transition) and the heads+loss pass (its own 4112 B per row) separately; `at_rollout_size` is
    # SURVEY 8(d)'s definition — one launch of each stand-alone kernel over T*N transitions, measured live.
    gae_s = ks["gae"]["total_s"] / ks["gae"]["launches"]
    in_run = dict(gae=dict(launch_s=gae_s, achieved=round(17.0 * transitions / gae_s / 1e9, 1), frac=round(17.0 * transitions / gae_s / HBM_PEAK, 4)))
    if "heads_loss_fwd_bwd" in ks:
        hl = ks["heads_loss_fwd_bwd"]
        in_run["heads_loss_fwd_bwd"] = dict(launch_s=hl["total_s"] / hl["launches"], achieved=kernels["heads_loss_fwd_bwd"]["achieved_GBps"],
                                            frac=kernels["heads_loss_fwd_bwd"]["frac"], note="loss inside the heads pass: 4112 B per row")
    elif "ppo_loss_fwd_bwd" in ks:
        loss_s = ks["ppo_loss_fwd_bwd"]["total_s"] / ks["ppo_loss_fwd_bwd"]["launches"] * cfg.num_minibatches
        in_run["ppo_loss_fwd_bwd"] = dict(launch_s=loss_s, achieved=round(56.0 * transitions / loss_s / 1e9, 1),
                                          frac=round(56.0 * transitions / loss_s / HBM_PEAK, 4), note="32 minibatch launches per pass")
    gae_loss = dict(kernel="gae(G1: chunk maps and — variant 3 — the carry pass composed inside the rollout launch; apply + moments launches) + ppo_loss_fwd_bwd",
                    in_run=in_run, at_rollout_size=full_pass(trainer, T, N, dev), bytes_per_pass=73.0 * transitions)
    roofline = dict(bound=head["bound"], achieved=head["achieved"], peak=head["peak"], unit=head["unit"], frac=head["frac"],
                    traffic=None, kernel=head["kernel"], launch_s=head["launch_s"],
                    traffic_from_profiles=(dict(file="profiles/" + prof[0], note="PMC byte counts of separate rocprofv3 --pmc passes over the "
                                                "same kernels of the SAME binary (sha256 of libgymrl_hip.so matches), not measured in this run",
                                                summary=prof[1]) if prof else prof_refused),
                    library_sha256=lib_hash,
                    gae_loss_pass=gae_loss, kernels=kernels)
    for k in ("flops_per_launch_group", "bytes_per_launch_group", "share_of_step"):
        if k in head:
            roofline[k] = head[k]
    if prof:
        # the counters of the same binary's six GEMM launches: HBM bytes per launch group beside the algorithmic ones, and the
        # clock the chip sustains under f32 MFMA (GRBM_GUI_ACTIVE / wall time; the 157.3 TFLOP/s peak assumes 2.4 GHz —
        # MI355X_MICROARCH.md "DVFS give-back": a dense matrix body clocks to its power budget, not to the nameplate)
        six = [prof[1][k] for k in ("gemm_fwd_256_tanh", "gemm_fwd_512", "gemm_dw_512", "gemm_dx_512_tanhbwd", "gemm_dw_256_db",
                                    "gemm_dx_256_tanhbwd") if k in prof[1]]
        if len(six) == 6:
            roofline["traffic"] = float(sum(e["hbm_bytes_per_launch"] for e in six))
            roofline["algorithmic_bytes_per_launch_group"] = float(sum(e["algorithmic_bytes_per_row"] for e in six) * prof[1]["rows"])
            t_us = sum(e["duration_us_profiled"] for e in six)
            clk = sum(e["effective_clock_GHz"] * e["duration_us_profiled"] for e in six) / t_us
            roofline["clock"] = dict(effective_GHz_under_counters=round(clk, 3), nameplate_GHz=2.4,
                                     peak_at_effective_clock=round(head["peak"] * clk / 2.4, 1),
                                     frac_at_effective_clock=round(head["frac"] * 2.4 / clk, 4),
                                     mfma_busy_share=round(sum(e["mfma_busy_share"] * e["duration_us_profiled"] for e in six) / t_us, 3))

    out = {
        "metric": "env-steps/sec at N envs/GPU (PPO LunarLander), 1/2/4/8 GPUs + %HBM roofline",
        "value": transitions * world * a.steps / dt,
        "unit": "env-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PPO LunarLander-v3, 4096 vectorised envs per MI355X (BASELINE.json configs[1])",
                   "envs_per_gpu": N, "rollout_len": T, "transitions_per_step_per_gpu": transitions,
                   "num_epochs": cfg.num_epochs, "num_minibatches": cfg.num_minibatches,
                   "minibatch": transitions // cfg.num_minibatches, "model": "ActorCritic 8-256-256-{256-4,256-1} tanh, 200,965 params",
                   "parallelism": f"dp{world} (env shards + flat-gradient all-reduce)" if world > 1 else "single GPU"},
        "roofline": roofline,
        "phases": {"rollout_ms": round(rollout_s * 1e3, 1), "update_ms": round(update_s * 1e3, 1),
                   "rollout_only_env_steps_per_s": round(transitions / rollout_s),
                   # one rollout under the initial (frozen) policy, outside the timed region: the in-run figure moves with
                   # what the policy has learned (contact mix), this one does not
                   "rollout_frozen_policy_ms": frozen_ms},
        "train_metrics": {k: float(v) for k, v in (metrics or {}).items()},
        "avg_episode_return": (sum(trainer.episode_rewards) / len(trainer.episode_rewards)) if trainer.episode_rewards else None,
        "comm": dict(comm_summary(world, trainer._reducer, ks, dt / a.steps, a.steps), backend=_backend_name(),
                     rank_devices=devices),
    }
    if world == 1 and not a.no_cpu_baseline:
        from oracle.ref_ppo_cpu import time_cpu_baseline
        out["cpu_baseline"] = time_cpu_baseline(a.cpu_budget)
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
