#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp6; mkdir -p $OUT
cd $ROOT
for lib in "" "$@"; do
  L=$ROOT/se3_equi_graph_registration_amd/libequigspr${lib}.so
  echo "== lib '$lib'"
  EQUIGSPR_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "33 or kitti or f33 or golden or c4 or random_graph or reproducible" 2>&1 | tail -3
  EQUIGSPR_LIB=$L python bench.py --no-cpu-baseline --workload c4 --steps 40 --warmup 5 > $OUT/c4${lib}.json 2>$OUT/c4${lib}.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/c4${lib}.json").read().strip().splitlines()[-1]); print("lib='$lib' c4 step_ms", d["ms_per_step"], "fwd", d["kernels"]["edge_fwd_ms"], "bwd", d["kernels"]["edge_bwd_ms"])
except Exception as ex: print("lib='$lib' FAILED", ex); print(open("$OUT/c4${lib}.err").read()[-600:])
PY
done
