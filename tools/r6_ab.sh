#!/bin/bash
O=gpurun_out/r6l; mkdir -p $O
PREV=$PWD/tools/probes/libgymrl_hip_prev.so
for rep in 1 2; do for v in new prev; do
  if [ $v = prev ]; then export GYMRL_HIP_LIB=$PREV; else unset GYMRL_HIP_LIB; fi
  python bench.py --algo ppo_full --steps 3 --warmup 1 --no-cpu-baseline > $O/ppo_full_${v}_$rep.json 2>/dev/null
done; done
unset GYMRL_HIP_LIB
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j=json.loads(line); print(sys.argv[1].split('/')[-1], round(j['value']/1e6,3),'M', round(j['ms_per_step'],1), j['phases'])
PY
done
