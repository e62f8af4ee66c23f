cd /root/repo
mkdir -p gpurun_out/c19
timeout 300 python bench.py --algo ppo_full --steps 3 --warmup 1 > gpurun_out/c19/pf.json 2> gpurun_out/c19/err.txt
python -c "
import json
d=json.load(open('gpurun_out/c19/pf.json')); print(round(d['value']/1e6,3), d['ms_per_step'], {k:v for k,v in d.items() if k in ('rollout_ms','update_ms')}, d.get('phases'))
"
