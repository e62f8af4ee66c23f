cd /root/repo
mkdir -p gpurun_out/c10
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_trainers_gpu.py tests/test_step_chunk_gpu.py tests/test_graphs_gpu.py -m gpu -q --maxfail=8 > gpurun_out/c10/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c10/pytest.txt
for f in "sac --batch 4096" "rainbow --batch 8192" "sac" "rainbow"; do n=$(echo $f | tr ' -' '__'); timeout 300 python bench.py --algo $f --steps 30 --warmup 5 > gpurun_out/c10/bench_$n.json 2> gpurun_out/c10/bench_$n.err; python -c "
import json
d=json.load(open('gpurun_out/c10/bench_$n.json')); print('$f', round(d['value']/1e6,2),'M', d['config']['ms_per_vector_step'], d['roofline']['frac'])
"; done
tail -4 gpurun_out/c10/pytest.txt
