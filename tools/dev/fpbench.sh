# dev: time the feature-parallel edge kernels (hidden_nf 64 / 48 / 16) and run their parity cases
set -e
for F in ${FP_WIDTHS:-64 48 16}; do
  python bench.py --feat $F --heads 2 --no-cpu-baseline --no-workloads --min-seconds 0.2 --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('F=$F ms/step',d['ms_per_step'], d['roofline']['kernel'][:20], 'fwd us', round(1e3*k['edge_fwd_ms'],2), 'bwd us', round(1e3*k['edge_bwd_ms'],2))
"
done
[ -n "$FP_NO_TESTS" ] || python -m pytest tests/test_gpu_parity.py -x -q -k "golden or random_graph or width or no_edge or partial" 2>&1 | tail -3
