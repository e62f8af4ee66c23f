#!/bin/bash
# Copies one tools/profile_round6.sh output directory (under gpurun_out/) into profiles/ under the r06_ names and
# rebuilds the PMC summary with its provenance stamp.  usage: tools/collect_round6.sh <dir under gpurun_out>
set -eu
cd "$(dirname "$0")/.."
R=gpurun_out/$1
for f in bench_final.json bench_20steps.json bench_ppo_full.json bench_sac.json bench_rainbow.json bench_sac_bigbatch.json \
         bench_rainbow_bigbatch.json bench_ppo_forced_rccl.json bench_ppo_full_forced_rccl.json bench_kernel_stats.csv bench_under_rocprof.json ppo_full_kernel_stats.csv sac_kernel_stats.csv \
         rainbow_kernel_stats.csv sac_bigbatch_kernel_stats.csv rainbow_bigbatch_kernel_stats.csv micro_per.txt micro_kernels.json rollout_balance.txt rainbow_timeline.txt sac_timeline.txt \
         pmc_FETCH_SIZE_counter_collection.csv pmc_WRITE_SIZE_counter_collection.csv; do
  cp "$R/$f" "profiles/r06_$f"
done
cp "$R/pmc_gemm_sq.csv" profiles/r06_pmc_gemm.csv
cp "$R/pmc_summary.json" profiles/r06_pmc_summary.json      # built on the GPU box before the bench lines ran (tools/profile_round6.sh)
python - <<'PY'
import json, hashlib
s = json.load(open("profiles/r06_pmc_summary.json"))["provenance"]
h = hashlib.sha256(open("gymrl_amd/libgymrl_hip.so", "rb").read()).hexdigest()
print("summary stamped with", s["libgymrl_hip_sha256"][:16], "| library in tree", h[:16], "|", "MATCH" if s["libgymrl_hip_sha256"] == h else "MISMATCH")
PY
