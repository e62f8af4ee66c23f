#!/bin/bash
O=gpurun_out/r6c; mkdir -p $O
R05=$PWD/tools/probes/libgymrl_hip_r05.so
for rep in 1 2; do
for v in cur r05; do
  if [ $v = r05 ]; then export GYMRL_HIP_LIB=$R05; else unset GYMRL_HIP_LIB; fi
  python bench.py --algo rainbow --no-cpu-baseline > $O/rainbow_${v}_$rep.json 2>/dev/null
  python bench.py --algo sac --no-cpu-baseline > $O/sac_${v}_$rep.json 2>/dev/null
done; done
unset GYMRL_HIP_LIB
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(j['value']/1e6,3),'M', j['config'].get('ms_per_vector_step'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
