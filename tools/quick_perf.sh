#!/bin/bash
# quick A/B numbers: edge kernel HIP-event times + step time for the headline, B=16, KITTI shard and the eval pair
#   gpurun -- 'bash tools/quick_perf.sh <tag>'      (EQUIGSPR_LIB selects a library variant)
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out
for w in "" "--workload c3" "--workload c4" "--workload c5"; do
  python3 $ROOT/bench.py --no-cpu-baseline --no-workloads --steps 50 --warmup 5 --min-seconds 0.3 $w 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$TAG [%s] %.4f ms/step  fwd %.2f us (%.3f)  bwd %s' % ('$w', d['ms_per_step'], k['edge_fwd_ms']*1e3, k['edge_fwd_hbm_frac'], ('%.2f us (%.3f)' % (k['edge_bwd_ms']*1e3, k['edge_bwd_hbm_frac'])) if 'edge_bwd_ms' in k else '-'))"
done | tee -a $ROOT/gpurun_out/quick_perf_$TAG.txt
