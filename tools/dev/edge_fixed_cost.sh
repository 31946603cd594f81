# dev: edge kernel launch time against the neighbour count of one 2 048-point pair -> fixed (per-launch) and per-tile cost
for K in 1 2 5 10 20 40; do
  python bench.py --k $K --no-cpu-baseline --no-workloads --no-fixed-pair --min-seconds 0.1 --steps 10 ${EXTRA_FLAGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('k=$K E', d['roofline']['edges_per_launch'], 'fwd us', round(1e3*k['edge_fwd_ms'],2), 'bwd us', round(1e3*k['edge_bwd_ms'],2))
"
done
