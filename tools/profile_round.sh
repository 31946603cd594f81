#!/bin/bash
# Evidence run for profiles/ (round 4 form; round 5 adds the GPU inference of BASELINE configs[0], its batched form and the typed
# irrep-basis network): for the headline and every other BASELINE workload -- the bench line, the rocprofv3
# kernel stats of the same command (hipGraph replays of the product loop), and the two PMC passes (FETCH_SIZE / WRITE_SIZE collected
# separately, kernel-trace only, eager launches) summarised per kernel (edge AND node kernels); then the default (driver-form) line,
# which quotes those files, and the launch sequence of one step.
#   gpurun --timeout 3000 -- 'bash tools/profile_round.sh r05_a'
# Outputs land in gpurun_out/<tag>/ ; tools/collect_profiles.sh copies what should be judged into profiles/.
set -u
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SPECS=""
one() {  # name, edges per launch, feat, training(0|1), bench args...
  local name=$1 edges=$2 feat=$3 training=$4; shift 4
  python3 "$ROOT/bench.py" --no-cpu-baseline --no-workloads "$@" > "$OUT/bench_$name.json" 2>> "$OUT/bench.err"
  rm -rf /tmp/prof_$name
  # (round 6: the per-kernel statistics are taken in the SERIAL form of the step -- EQUIGSPR_BUILD_AHEAD=0, the graph build inside the step
  # graph: the same kernels, one queue, no overlap -- so that a launch's duration is its own; the shipped form runs the build on a second
  # stream beside the previous step, where durations of overlapping launches inflate each other and one cold replay of the build graph
  # dominates the averages.  The concurrent form's timeline is recorded separately: step_sequence.txt / gaps.txt below.)
  EQUIGSPR_BUILD_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-workloads --min-seconds 0 "$@" > "$OUT/stats_$name.log" 2>&1
  cp "$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_$name.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_${name}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_${name}_$c -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-workloads --no-graph --min-seconds 0 "$@" --steps 5 --warmup 2 > "$OUT/pmc_${name}_$c.log" 2>&1
    cp "$(find /tmp/prof_${name}_$c -name '*counter_collection.csv' | head -1)" "$OUT/${name}_${c}_counter_collection.csv"
  done
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/${name}_FETCH_SIZE_counter_collection.csv" "$OUT/${name}_WRITE_SIZE_counter_collection.csv" "$OUT/pmc_$name.json" $edges $feat "$TAG" "$*" > /dev/null
  local flags="$*"
  SPECS="$SPECS $OUT/kernel_stats_$name.csv:$edges:$feat:$training:${flags// /+}"
  tail -1 "$OUT/bench_$name.json" | cut -c1-160
}
one b1 81920 32 1 --steps 200 --warmup 20
one b16 1310720 32 1 --workload c3 --steps 60 --warmup 10
one c4shard 1310720 33 1 --workload c4 --steps 40 --warmup 5
one c5 524288 32 0 --workload c5 --steps 100 --warmup 10
one b16_irreps 1310720 32 1 --workload c3 --edge-basis irreps --steps 60 --warmup 10
one c1infer 81920 32 0 --forward-only --eval-head --steps 300 --warmup 20
one c1infer_b16 1310720 32 0 --forward-only --eval-head --batch 16 --steps 40 --warmup 5
one irrepgnn 81920 32 1 --network irreps --steps 60 --warmup 10
one irrepgnn_b8 655360 32 1 --network irreps --batch 8 --steps 20 --warmup 3
python3 "$ROOT/tools/in_step_summary.py" "$OUT/in_step.json" "$TAG" $SPECS
# the default line quotes the evidence above when it sits in profiles/: place this run's files there for the run (on the GPU box only)
cp "$OUT/in_step.json" "$ROOT/profiles/${TAG}_in_step.json"
for w in b1 b16 b16_irreps c4shard c5 c1infer c1infer_b16 irrepgnn irrepgnn_b8; do cp "$OUT/pmc_$w.json" "$ROOT/profiles/${TAG}_pmc_$w.json"; done
python3 "$ROOT/bench.py" > "$OUT/bench.json" 2>> "$OUT/bench.err"
cp "$ROOT/bench_full.json" "$OUT/bench_full.json"
tail -1 "$OUT/bench.json" | cut -c1-300
EQUIGSPR_BUILD_AHEAD=0 bash "$ROOT/tools/step_sequence.sh" > "$OUT/step_sequence.txt" 2>&1
# the shipped (build-ahead) form: both queues of two consecutive steps, and the idle gap between step graphs
rm -rf /tmp/seq_ahead
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq_ahead -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-workloads --no-fixed-pair --min-seconds 0 --steps 60 --warmup 10 > /tmp/seq_ahead.log 2>&1
python3 "$ROOT/tools/dev/trace_gaps2.py" "$(find /tmp/seq_ahead -name '*kernel_trace.csv' | head -1)" --timeline > "$OUT/step_sequence_build_ahead.txt" 2>&1
ls -la "$OUT"
