#!/bin/bash
O=gpurun_out/r6n; mkdir -p $O
P=$PWD/tools/probes/libgymrl_hip_base.so
python -m pytest tests -m gpu -x -q -k "rollout or ppo or trainer or gae or run_to_run" 2>&1 | tail -2
for rep in 1 2 3; do for v in base prio; do
  if [ $v = base ]; then export GYMRL_HIP_LIB=$P; else unset GYMRL_HIP_LIB; fi
  python bench.py --algo ppo_full --steps 3 --warmup 1 --no-cpu-baseline > $O/full_${v}_$rep.json 2>/dev/null
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/ppo_${v}_$rep.json 2>/dev/null
done; done
unset GYMRL_HIP_LIB
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j=json.loads(line); print(sys.argv[1].split('/')[-1], round(j['value']/1e6,3),'M', round(j['ms_per_step'],1), j['phases'].get('rollout_ms'), j['phases'].get('update_ms'))
PY
done
