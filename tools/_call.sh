cd /root/repo
mkdir -p gpurun_out/c9
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k "clip_adam or adam" > gpurun_out/c9/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c9/pytest.txt
timeout 300 python bench.py --algo rainbow > gpurun_out/c9/bench_rainbow.json 2> gpurun_out/c9/bench_rainbow.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c9/bench3.json 2> gpurun_out/c9/bench3.err
tail -3 gpurun_out/c9/pytest.txt
python -c "
import json
d=json.load(open('gpurun_out/c9/bench_rainbow.json')); print('rainbow', round(d['value']/1e6,2),'M', d['config']['ms_per_vector_step'])
d=json.load(open('gpurun_out/c9/bench3.json')); print(round(d['value']/1e6,3),'M', d['ms_per_step'], d['roofline']['frac'])
k=d['roofline'].get('kernels',{})
print(sum(v.get('avg_us',0) for n,v in k.items() if n not in ('rollout_chunk','gae')), {n:v.get('avg_us') for n,v in k.items() if n in ('adam_step','update_finalize','gather_minibatch')})
"
