cd /root/repo
mkdir -p gpurun_out/c15
timeout 600 python -m pytest tests/test_hip_parity_offpolicy.py tests/test_fused_step_gpu.py -m gpu -q --maxfail=8 -k "sumtree or per_ or rainbow" > gpurun_out/c15/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c15/pytest.txt
timeout 200 python tools/micro_per.py > gpurun_out/c15/micro_per.txt 2>&1
timeout 300 python bench.py --algo rainbow --batch 8192 --steps 30 --warmup 5 > gpurun_out/c15/bench_rb_big.json 2> gpurun_out/c15/err.txt
tail -4 gpurun_out/c15/pytest.txt; grep "1024\|512 idx" gpurun_out/c15/micro_per.txt
python -c "
import json
d=json.load(open('gpurun_out/c15/bench_rb_big.json')); print('rainbow big', round(d['value']/1e6,2),'M', d['config']['ms_per_vector_step'], d['roofline']['frac'], [(k[:30],v.get('us')) for k,v in d['pieces'].items() if 'per_update B' in k])
"
