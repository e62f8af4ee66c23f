#!/bin/bash
# Runs on the GPU box (through tools/gpu.sh): the round-6 measurement set committed under profiles/.
# usage: tools/profile_round6.sh <outdir under gpurun_out>
set -u
OUT=/root/repo/gpurun_out/$1
mkdir -p "$OUT"
cd /root/repo
# PMC passes first: the summary they produce is stamped with this library's sha256, and bench.py attaches `traffic` / `clock`
# to the lines below only from a summary whose stamp matches the library it loads
pushd /tmp > /dev/null && export TMPDIR=/tmp
export GYMRL_PMC_PROVENANCE="$OUT/pmc_provenance.json"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -- python /root/repo/tools/pmc_gemm.py > /dev/null 2>&1
  cp "$(find /tmp/p_$c -name '*counter_collection.csv' | head -1)" "$OUT/pmc_${c}_counter_collection.csv"
done
rm -rf /tmp/p_sq
timeout -k 5 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p_sq -- python /root/repo/tools/pmc_gemm.py > /dev/null 2>&1
cp "$(find /tmp/p_sq -name '*counter_collection.csv' | head -1)" "$OUT/pmc_gemm_sq.csv"
cp "$(find /tmp/p_sq -name '*kernel_trace.csv' | head -1)" "$OUT/pmc_gemm_sq_trace.csv"
python /root/repo/tools/pmc_gemm_summarise.py "$OUT/pmc_FETCH_SIZE_counter_collection.csv" "$OUT/pmc_WRITE_SIZE_counter_collection.csv" \
  "$OUT/pmc_gemm_sq.csv" "$OUT/pmc_gemm_sq_trace.csv" /root/repo/profiles/r06_pmc_summary.json "$OUT/pmc_provenance.json" > /dev/null
cp /root/repo/profiles/r06_pmc_summary.json "$OUT/pmc_summary.json"
popd > /dev/null
timeout 600 python bench.py > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_20steps.json" 2> /dev/null
timeout 300 python bench.py --algo ppo_full --steps 3 --warmup 1 > "$OUT/bench_ppo_full.json" 2> /dev/null
timeout 300 python bench.py --algo sac > "$OUT/bench_sac.json" 2> /dev/null
timeout 300 python bench.py --algo rainbow > "$OUT/bench_rainbow.json" 2> /dev/null
timeout 300 python bench.py --algo sac --batch 4096 --steps 30 --warmup 30 > "$OUT/bench_sac_bigbatch.json" 2> /dev/null
timeout 300 python bench.py --algo rainbow --batch 8192 --steps 30 --warmup 30 > "$OUT/bench_rainbow_bigbatch.json" 2> /dev/null
# the multi-GPU path on RCCL with ONE rank (dist.force_collectives): the driver's launch shape, every collective issued
GYMRL_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 \
  bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_ppo_forced_rccl.json" 2> /dev/null
GYMRL_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 \
  bench.py --algo ppo_full --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_ppo_full_forced_rccl.json" 2> /dev/null
timeout 200 python tools/micro_kernels.py > "$OUT/micro_kernels.json" 2> /dev/null
timeout 200 python tools/micro_per.py > "$OUT/micro_per.txt" 2> /dev/null
timeout 200 python tools/probe_rollout_balance.py 2048 > "$OUT/rollout_balance.txt" 2> /dev/null
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <name> <cmd...>: kernel stats CSV of a command
  local name=$1; shift
  rm -rf /tmp/p_$name
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- "$@" > "$OUT/${name}_under_rocprof.json" 2> /dev/null
  cp "$(find /tmp/p_$name -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv"
}
prof bench python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline
prof ppo_full python /root/repo/bench.py --algo ppo_full --rollout 256 --steps 1 --warmup 1
prof rainbow python /root/repo/bench.py --algo rainbow --steps 10 --warmup 2
prof sac python /root/repo/bench.py --algo sac --steps 10 --warmup 2
# two vector steps of each off-policy chunk on the queue timeline (tools/trace_steps.py)
for a in rainbow:rainbow_act_kernel sac:sac_act_kernel; do
  rm -rf /tmp/t_${a%%:*}
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_${a%%:*} -- python /root/repo/bench.py --algo ${a%%:*} --steps 64 --warmup 32 > /dev/null 2>&1
  python /root/repo/tools/trace_steps.py "$(find /tmp/t_${a%%:*} -name '*kernel_trace.csv' | head -1)" ${a##*:} 2 > "$OUT/${a%%:*}_timeline.txt"
done
prof sac_bigbatch python /root/repo/bench.py --algo sac --batch 4096 --steps 4 --warmup 1
prof rainbow_bigbatch python /root/repo/bench.py --algo rainbow --batch 8192 --steps 4 --warmup 1
ls -la "$OUT"
for f in final 20steps ppo_full sac rainbow sac_bigbatch rainbow_bigbatch; do python -c "
import json
try:
    d=json.load(open('$OUT/bench_$f.json')); print('$f', round(d['value']/1e6,3), 'M', round(d['ms_per_step'],3), 'ms', d['roofline']['frac'])
except Exception as e: print('$f FAILED', e)
"; done
