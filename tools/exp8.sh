#!/bin/bash
# kernel stats per library variant: exp8.sh <suffix...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  [ "$lib" = base ] && lib=""
  export EQUIGSPR_LIB=$ROOT/se3_equi_graph_registration_amd/libequigspr${lib}.so
  bash $ROOT/tools/kstat.sh "b1$lib" --batch 1 --steps 60 --warmup 5 | grep -E "^==|node_bwd16"
  bash $ROOT/tools/kstat.sh "b16$lib" --batch 16 --steps 20 --warmup 3 | grep -E "^==|node_bwd16"
done
