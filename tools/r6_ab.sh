#!/bin/bash
O=gpurun_out/r6i; mkdir -p $O
rm -f $O/ledger.jsonl
GYMRL_TOL_LEDGER=$PWD/$O/ledger.jsonl python -m pytest tests -m gpu -x -q 2>&1 | tail -2
{
echo "# End-to-end learning sanity on round 6's final library (one MI355X)."
echo "# PPO LunarLander-v3: python tools/try_ppo.py LunarLander-v3 4096 512 60"
python tools/try_ppo.py LunarLander-v3 4096 512 60 2>&1 | grep -v amdgpu.ids | awk 'NR<=5 || /it (0|10|20|30|40|50|59) / || /Eval|eval/'
echo; echo "# Rainbow CartPole-v1 on the chunked fused step: python tools/try_offpolicy.py rainbow 16 40000"
python tools/try_offpolicy.py rainbow 16 40000 2>&1 | grep -v amdgpu.ids
echo; echo "# SAC Pendulum-v1 on the fused step: python tools/try_offpolicy.py sac 16 32000"
python tools/try_offpolicy.py sac 16 32000 2>&1 | grep -v amdgpu.ids
} > $O/learning_sanity.txt
tail -5 $O/learning_sanity.txt
