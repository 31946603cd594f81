#!/bin/bash
# dev experiment: parity + timing of the 16-edge kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp1
mkdir -p $OUT
cd $ROOT
export EQUIGSPR_EDGE_SAVE_MB=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -25 > $OUT/pytest_new.txt
tail -5 $OUT/pytest_new.txt
for chunk in 16 32 64; do
  EQUIGSPR_EDGE_CHUNK=$chunk python bench.py --no-cpu-baseline --steps 300 --warmup 30 > $OUT/b1_chunk$chunk.json 2>$OUT/b1_chunk$chunk.err
  python - <<PY
import json; d=json.loads(open("$OUT/b1_chunk$chunk.json").read().strip().splitlines()[-1]); print("B1 chunk$chunk", d["ms_per_step"], d["kernels"]["edge_fwd_ms"], d["kernels"]["edge_bwd_ms"])
PY
  EQUIGSPR_EDGE_CHUNK=$chunk python bench.py --no-cpu-baseline --batch 16 --steps 60 --warmup 10 > $OUT/b16_chunk$chunk.json 2>$OUT/b16_chunk$chunk.err
  python - <<PY
import json; d=json.loads(open("$OUT/b16_chunk$chunk.json").read().strip().splitlines()[-1]); print("B16 chunk$chunk", d["ms_per_step"], d["kernels"]["edge_fwd_ms"], d["kernels"]["edge_bwd_ms"])
PY
done
unset EQUIGSPR_EDGE_SAVE_MB
python bench.py --no-cpu-baseline --steps 300 --warmup 30 > $OUT/b1_old.json 2>$OUT/b1_old.err
python - <<PY
import json; d=json.loads(open("$OUT/b1_old.json").read().strip().splitlines()[-1]); print("B1 old(save)", d["ms_per_step"], d["kernels"]["edge_fwd_ms"], d["kernels"]["edge_bwd_ms"])
PY
