#!/bin/bash
O=gpurun_out/r6d; mkdir -p $O
# the driver's multi-GPU launch shape with ONE process, collectives forced through RCCL
GYMRL_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 \
  bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/torchrun_ppo.json 2> $O/torchrun_ppo.err; echo "torchrun ppo rc=$?"
GYMRL_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 \
  bench.py --algo ppo_full --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/torchrun_ppo_full.json 2> $O/torchrun_ppo_full.err; echo "torchrun ppo_full rc=$?"
python bench.py --algo ppo_full --steps 2 --warmup 1 --no-cpu-baseline > $O/ppo_full.json 2> $O/ppo_full.err; echo "ppo_full rc=$?"
python bench.py --gpus 1 --spawn-selftest > $O/selftest.json 2>&1; echo "selftest rc=$?"
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j=json.loads(line)
        if 'value' in j: print(sys.argv[1].split('/')[-1], round(j['value']/1e6,3),'M', round(j['ms_per_step'],2), j.get('phases'), (j.get('comm') or {}).get('grad_allreduce'))
        else: print(sys.argv[1].split('/')[-1], j)
PY
done
