#!/bin/bash
# B=1 step and edge kernel times per forced edge chunk size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for ck in 0 16 32 48 64; do
  EQUIGSPR_EDGE_CHUNK=$ck python bench.py --no-cpu-baseline --steps 300 --warmup 20 --min-seconds 0.5 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ck step_ms', d['ms_per_step'], 'fwd', d['kernels']['edge_fwd_ms'], 'bwd', d['kernels']['edge_bwd_ms'])"
done
