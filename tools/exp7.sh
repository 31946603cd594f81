#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_irreps.py -m gpu -q -x 2>&1 | tail -3
bash tools/exp5.sh exp7 "$@"
