#!/bin/bash
# Runs on the GPU box (through tools/gpu.sh): the measurement set committed under profiles/ each round.
# usage: tools/profile_round.sh <outdir under gpurun_out>
set -u
OUT=/root/repo/gpurun_out/$1
mkdir -p "$OUT"
cd /root/repo
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 200 python tools/micro_kernels.py > "$OUT/micro_kernels.json" 2> /dev/null
timeout 200 python tools/micro_mlp.py > "$OUT/micro_mlp.json" 2> /dev/null
timeout 200 python tools/micro_update.py > "$OUT/micro_update.json" 2> /dev/null
timeout 300 python tools/micro_rollout.py 512 > "$OUT/micro_rollout.json" 2> /dev/null
timeout 300 python tools/micro_offpolicy.py > "$OUT/micro_offpolicy.json" 2> /dev/null
timeout 200 python tools/probe_rollout_balance.py 2048 > "$OUT/rollout_balance.txt" 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> /dev/null
cp "$(find /tmp/p_bench -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -- python /root/repo/tools/pmc_kernels.py > /dev/null 2>&1
  cp "$(find /tmp/p_$c -name '*counter_collection.csv' | head -1)" "$OUT/pmc_${c}_counter_collection.csv"
done
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
  --output-format csv -d /tmp/p_sq -- python /root/repo/tools/probe_rollout_balance.py 512 > /dev/null 2>&1
grep -E "Counter_Name|rollout_lunar" "$(find /tmp/p_sq -name '*counter_collection.csv' | head -1)" > "$OUT/pmc_rollout_sq.csv"
grep -E "Kernel_Name|rollout_lunar" "$(find /tmp/p_sq -name '*kernel_trace.csv' | head -1)" | cut -d, -f1-12 > "$OUT/pmc_rollout_trace.csv"
ls -la "$OUT"
tail -c 600 "$OUT/bench.json"
