#!/bin/bash
# Same-box A/B of two builds of the library through bench.py (how every "A/B in one box" figure of round 6 was taken):
#   tools/r6_ab.sh <probe .so (absolute path on the box, e.g. $PWD/tools/probes/libgymrl_hip_x.so)> <runs> <bench.py arguments ...>
# alternates the in-tree library ("new") and the probe library ("probe", through GYMRL_HIP_LIB) <runs> times in one process chain on
# ONE box and prints value, ms per step and the phases of every run.  Build the probe library beside the product, e.g. from the
# previous commit:  git archive HEAD~1 gymrl_amd/csrc include | tar -x -C /tmp/prev && make -C /tmp/prev/gymrl_amd/csrc -j8 &&
# cp /tmp/prev/gymrl_amd/libgymrl_hip.so tools/probes/libgymrl_hip_prev.so   (tools/probes/*.so is git-ignored and travels with gpurun).
PROBE=$1; RUNS=$2; shift 2
O=gpurun_out/ab_$$; mkdir -p $O
for rep in $(seq 1 "$RUNS"); do for v in new probe; do
  if [ $v = probe ]; then export GYMRL_HIP_LIB=$PROBE; else unset GYMRL_HIP_LIB; fi
  python bench.py "$@" --no-cpu-baseline > $O/${v}_$rep.json 2>/dev/null
done; done
unset GYMRL_HIP_LIB
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j = json.loads(line)
        print(sys.argv[1].split('/')[-1], round(j['value'] / 1e6, 3), 'M', round(j['ms_per_step'], 2), j.get('phases') or j['config'].get('ms_per_vector_step'))
PY
done
