#!/bin/bash
# quick timing: exp4.sh <tag> ; B=1 chunk 16/32, B=16 chunk 64, B=4 auto
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
export EQUIGSPR_EDGE_SAVE_MB=0
one() {  # lib chunk batch steps
  local lib=$1 chunk=$2 b=$3 steps=$4
  local L=$ROOT/se3_equi_graph_registration_amd/libequigspr${lib}.so
  EQUIGSPR_LIB=$L EQUIGSPR_EDGE_CHUNK=$chunk python bench.py --no-cpu-baseline --batch $b --steps $steps --warmup 20 > $OUT/b${b}_c${chunk}${lib}.json 2>$OUT/b${b}_c${chunk}${lib}.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b${b}_c${chunk}${lib}.json").read().strip().splitlines()[-1]); print("lib='$lib' B=$b chunk=$chunk step_ms", d["ms_per_step"], "fwd", d["kernels"]["edge_fwd_ms"], "bwd", d["kernels"]["edge_bwd_ms"])
except Exception as ex: print("lib='$lib' B=$b chunk=$chunk FAILED", ex)
PY
}
for lib in "" "$@"; do
  one "$lib" 16 1 300; one "$lib" 32 1 300; one "$lib" 0 4 100; one "$lib" 64 16 60
done
