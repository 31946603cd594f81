#!/bin/bash
# dev experiment: parity + timing of kernel variants.  usage: exp3.sh <tag> [lib suffixes...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
export EQUIGSPR_EDGE_SAVE_MB=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q --deselect "tests/test_gpu_parity.py::test_saved_activation_backward_equals_recompute_backward" 2>&1 | tail -8 > $OUT/pytest.txt
tail -4 $OUT/pytest.txt
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
  for chunk in 16 32; do one "$lib" $chunk 1 300; done
  for chunk in 32 64; do one "$lib" $chunk 16 60; done
done
