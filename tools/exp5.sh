#!/bin/bash
# A/B step time of library variants: exp5.sh <tag> <suffix...>   (B=1 default bench and B=16)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
for lib in "" "$@"; do
  L=$ROOT/se3_equi_graph_registration_amd/libequigspr${lib}.so
  for b in 1 16; do
    steps=300; [ $b = 16 ] && steps=60
    EQUIGSPR_LIB=$L python bench.py --no-cpu-baseline --batch $b --steps $steps --warmup 20 > $OUT/b${b}${lib}.json 2>$OUT/b${b}${lib}.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/b${b}${lib}.json").read().strip().splitlines()[-1]); print("lib='$lib' B=$b step_ms", d["ms_per_step"], "min", d["timing"]["ms_per_step_min"], "fwd", d["kernels"]["edge_fwd_ms"], "bwd", d["kernels"]["edge_bwd_ms"])
except Exception as ex: print("lib='$lib' B=$b FAILED", ex)
PY
  done
done
