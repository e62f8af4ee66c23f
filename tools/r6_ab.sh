#!/bin/bash
O=gpurun_out/r6j_$1; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/ppo20.json 2>/dev/null
python - $O/ppo20.json <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j=json.loads(line); r=j['roofline']
        print(round(j['value']/1e6,3),'M', round(j['ms_per_step'],1), j['phases']['rollout_ms'], j['phases']['update_ms'], 'frac', r['frac'], 'gae', r['gae_loss_pass']['in_run']['gae']['frac'], r['gae_loss_pass']['at_rollout_size']['frac'])
PY
