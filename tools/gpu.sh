#!/bin/bash
# Rebuild both libraries, then run a command on the MI355X box through gpurun.
# usage: tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C gymrl_amd/csrc -s -j8 2>&1 | grep -E "error|Error" && exit 1
make -C oracle -s 2>&1 | grep -E "error" && exit 1
mkdir -p gpurun_out
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
