# dev: A/B of library variants on the step time: ab_step.sh "<variant names, '' = shipped>" <bench args...>
VARS=$1; shift
for v in $VARS; do
  if [ "$v" != "base" ]; then export EQUIGSPR_LIB=$PWD/se3_equi_graph_registration_amd/libequigspr_$v.so; else unset EQUIGSPR_LIB; fi
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-workloads "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d.get('kernels',{})
print('$v', 'ms/step', d['ms_per_step'], 'fixed', (d.get('fixed_pair_replay') or {}).get('ms_per_step'), 'edge fwd/bwd us', round(1e3*k.get('edge_fwd_ms',0),2), round(1e3*k.get('edge_bwd_ms',0),2))
"
  done
done
