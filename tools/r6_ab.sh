#!/bin/bash
O=gpurun_out/r6g; mkdir -p $O
TK1=$PWD/tools/probes/libgymrl_hip_tk1.so
python -m pytest tests/test_fused_step_gpu.py tests/test_run_to_run_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in queue tk1; do
  if [ $v = tk1 ]; then export GYMRL_HIP_LIB=$TK1; else unset GYMRL_HIP_LIB; fi
  python bench.py --algo sac --batch 4096 --no-cpu-baseline > $O/sac_big_${v}_$rep.json 2>/dev/null
  python bench.py --algo rainbow --batch 8192 --no-cpu-baseline > $O/rainbow_big_${v}_$rep.json 2>/dev/null
  python bench.py --algo sac --batch 1024 --no-cpu-baseline > $O/sac_1k_${v}_$rep.json 2>/dev/null
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
