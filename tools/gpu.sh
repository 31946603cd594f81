#!/bin/bash
# build, then run a command on the GPU box:   bash tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C se3_equi_graph_registration_amd/csrc -j8 2>&1 | grep -E "error|warning: unused|Error" || true
T=$1; shift
/usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
