#!/bin/bash
# Evidence run for profiles/: the default bench line, the B=16 / KITTI-shard / eval-sweep lines, the rocprofv3 kernel stats of
# each of those commands, the two PMC passes (FETCH_SIZE / WRITE_SIZE collected separately, kernel-trace only) and the launch
# sequence of one step.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02_v1'
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | cut -c1-300
stats() {  # name, bench args...
  local name=$1; shift
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-workloads "$@" > "$OUT/bench_$name.json" 2>> "$OUT/bench.err"
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-workloads --min-seconds 0 "$@" > "$OUT/stats_$name.log" 2>&1
  cp "$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_$name.csv"
  tail -1 "$OUT/bench_$name.json" | cut -c1-200
}
stats b1 --steps 200 --warmup 20
stats b16 --workload c3 --steps 60 --warmup 10
stats c4shard --workload c4 --steps 40 --warmup 5
stats c5 --workload c5 --steps 100 --warmup 10
stats b16_irreps --workload c3 --edge-basis irreps --steps 60 --warmup 10
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-workloads --no-graph --min-seconds 0 > "$OUT/pmc_$c.log" 2>&1
  cp "$(find /tmp/prof_$c -name '*counter_collection.csv' | head -1)" "$OUT/${c}_counter_collection.csv"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/FETCH_SIZE_counter_collection.csv" "$OUT/WRITE_SIZE_counter_collection.csv" "$OUT/pmc_edge_kernels.json" 81920 32 "$TAG" > /dev/null
bash "$ROOT/tools/step_sequence.sh" > "$OUT/step_sequence.txt" 2>&1
ls -la "$OUT"
