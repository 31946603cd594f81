#!/bin/bash
# Evidence run for profiles/: the default bench line, the rocprofv3 kernel stats of the same command, the two PMC passes
# (FETCH_SIZE / WRITE_SIZE collected separately, kernel-trace only) and the launch sequence of one step.
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh v25'
# Outputs land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | cut -c1-400
python3 "$ROOT/bench.py" --batch 16 --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_b16.json" 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/stats.log" 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-graph > "$OUT/pmc_$c.log" 2>&1
  cp "$(find /tmp/prof_$c -name '*counter_collection.csv' | head -1)" "$OUT/${c}_counter_collection.csv"
done
bash "$ROOT/tools/step_sequence.sh" > "$OUT/step_sequence.txt" 2>&1
ls -la "$OUT"
