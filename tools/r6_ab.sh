#!/bin/bash
V4=$PWD/tools/probes/libgymrl_hip_gaev4.so
for rep in 1 2; do
for v in v2 v4; do
  if [ $v = v4 ]; then export GYMRL_HIP_LIB=$V4; else unset GYMRL_HIP_LIB; fi
  python tools/micro_kernels.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$v $rep', {k:(round(v['us'],1), round(v.get('frac',0),3)) for k,v in j.items() if k.startswith('gae') or k.startswith('copy')})"
done; done
unset GYMRL_HIP_LIB
GYMRL_HIP_LIB=$V4 python -m pytest tests/test_hip_parity.py -q -k "gae or moments" 2>&1 | grep -E "^E  |FAILED|Error" | head -20
