cd /root/repo
mkdir -p gpurun_out/c14
for c in 16 32 16 32 16 32; do for a in sac rainbow; do GYMRL_CHUNK=$c timeout 300 python bench.py --algo $a > gpurun_out/c14/b.json 2> /dev/null; python -c "
import json
d=json.load(open('gpurun_out/c14/b.json')); print('$a chunk $c', round(d['value']/1e6,2),'M', d['config']['ms_per_vector_step'])
"; done; done
