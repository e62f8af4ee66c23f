cd /root/repo
mkdir -p gpurun_out/c1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/c1/pytest.txt
timeout 300 python tools/probe_rollout_placement.py 2048 0 > gpurun_out/c1/placement_it0.txt 2>&1
timeout 300 python tools/probe_rollout_placement.py 2048 8 > gpurun_out/c1/placement_it8.txt 2>&1
timeout 200 python tools/probe_rollout_balance.py 2048 > gpurun_out/c1/balance.txt 2>&1
tail -5 gpurun_out/c1/pytest.txt; cat gpurun_out/c1/placement_it0.txt
