#!/bin/bash
O=gpurun_out/r6h; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "rollout or gae or ppo or trainer or multirank or run_to_run or abi or smoke" 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ppo_a.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ppo_b.json 2>/dev/null
for f in $O/ppo_*.json; do python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j=json.loads(line)
        g=j['roofline']['gae_loss_pass']['in_run']['gae']
        print(sys.argv[1].split('/')[-1], round(j['value']/1e6,3),'M', round(j['ms_per_step'],2), j.get('phases'), 'gae', round(g['launch_s']*1e6,1), g['frac'], j['roofline']['kernels']['gae'])
PY
done
