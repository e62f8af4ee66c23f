#!/bin/bash
# The 1 -> 8 GPU scaling table in one command, for the day an 8-GPU MI355X node is available (BASELINE configs[1] and
# configs[4]: PPO and PPO-full LunarLander, 4096 envs per GPU, RCCL all-reduce of the flat gradient over xGMI).
#   tools/scale.sh [steps] [warmup]           # on the node itself, from the repo root
# bench.py --gpus N spawns and verifies its own N ranks (one process per GPU, 127.0.0.1 rendezvous, exit 2 if the
# communicator is not exactly N RCCL ranks).  Prints one row per (algo, N): whole-job env-steps/s, ms per step, scaling
# efficiency against N = 1, and the exposed share of the gradient all-reduce in the step (comm.grad_allreduce).
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-5}; WARM=${2:-2}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${SCALE_OUT:-gpurun_out/scale}
mkdir -p "$OUT"
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
for algo in ppo ppo_full; do
  for n in 1 2 4 8; do
    [ "$n" -le "$NG" ] || { echo "[scale] skipping $algo x$n: only $NG device(s)"; continue; }
    timeout 1200 python bench.py --algo $algo --gpus $n --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline \
      > "$OUT/${algo}_n${n}.json" 2> "$OUT/${algo}_n${n}.err" || echo "[scale] $algo x$n failed (rc $?): see $OUT/${algo}_n${n}.err"
  done
done
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
print(f"{'algo':9s} {'gpus':>4s} {'env-steps/s':>14s} {'ms/step':>10s} {'efficiency':>10s} {'allreduce exposed % of step':>28s}")
for algo in ("ppo", "ppo_full"):
    base = None
    for n in (1, 2, 4, 8):
        p = os.path.join(out, f"{algo}_n{n}.json")
        try:
            line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
        except Exception:
            continue
        if d.get("n_gpus") != n:
            print(f"{algo:9s} {n:4d}  line reports n_gpus = {d.get('n_gpus')}: ignored")
            continue
        base = base or d["value"] / d["n_gpus"]
        comm = ((d.get("comm") or {}).get("grad_allreduce") or {})
        exposed = comm.get("exposed_pct_of_step")
        print(f"{algo:9s} {n:4d} {d['value']:14.0f} {d['ms_per_step']:10.1f} {d['value'] / (n * base):10.3f} "
              f"{'' if exposed is None else format(exposed, '28.3f')}")
PY
