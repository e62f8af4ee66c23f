#!/bin/bash
# round 6, first GPU call: the GPU suite, the headline line, the large-batch lines on the ticketed row kernels, one forced-RCCL line
mkdir -p gpurun_out/r6a
O=gpurun_out/r6a
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 3 --warmup 1 > $O/bench_ppo.json 2> $O/bench_ppo.err; echo "bench rc=$?"
python bench.py --algo sac --batch 4096 --no-cpu-baseline > $O/bench_sac_big.json 2> $O/bench_sac_big.err; echo "sac big rc=$?"
python bench.py --algo rainbow --batch 8192 --no-cpu-baseline > $O/bench_rainbow_big.json 2> $O/bench_rainbow_big.err; echo "rainbow big rc=$?"
python bench.py --algo sac --no-cpu-baseline > $O/bench_sac.json 2> $O/bench_sac.err; echo "sac rc=$?"
python bench.py --algo rainbow --no-cpu-baseline > $O/bench_rainbow.json 2> $O/bench_rainbow.err; echo "rainbow rc=$?"
GYMRL_FORCE_COLLECTIVES=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 \
  python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ppo_rccl1.json 2> $O/bench_ppo_rccl1.err; echo "forced rccl rc=$?"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(j['value']/1e6,3),'M', round(j['ms_per_step'],3),'ms', j.get('phases'), j.get('comm',{}).get('grad_allreduce'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
