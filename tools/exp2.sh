#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp2; mkdir -p $OUT
cd $ROOT
make -C se3_equi_graph_registration_amd/csrc trace 2>&1 | tail -2
export EQUIGSPR_EDGE_SAVE_MB=0
for B in 1 16; do
  python tools/trace_edge16.py $B > $OUT/trace_b$B.txt 2>&1; cat $OUT/trace_b$B.txt
done
