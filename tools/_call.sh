cd /root/repo
mkdir -p gpurun_out/c4
timeout 600 python -m pytest tests/test_hip_parity_offpolicy.py tests/test_fused_step_gpu.py -m gpu -q --maxfail=8 -k "sumtree or per_ or rainbow" > gpurun_out/c4/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c4/pytest.txt
timeout 200 python tools/micro_per.py > gpurun_out/c4/micro_per.txt 2>&1
timeout 300 python bench.py --algo sac > gpurun_out/c4/bench_sac.json 2> gpurun_out/c4/bench_sac.err
timeout 300 python bench.py --algo rainbow > gpurun_out/c4/bench_rainbow.json 2> gpurun_out/c4/bench_rainbow.err
tail -4 gpurun_out/c4/pytest.txt; cat gpurun_out/c4/micro_per.txt
for f in rainbow sac; do python -c "
import json
d=json.load(open('gpurun_out/c4/bench_$f.json')); print('$f', round(d['value']/1e6,2),'M', d['config']['ms_per_vector_step'])
"; done
