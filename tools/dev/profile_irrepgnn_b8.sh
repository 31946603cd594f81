set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r06_b; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
name=irrepgnn_b8; ARGS="--network irreps --batch 8 --steps 20 --warmup 3"
python3 $ROOT/bench.py --no-cpu-baseline --no-workloads $ARGS > $OUT/bench_$name.json 2>> $OUT/bench.err
EQUIGSPR_BUILD_AHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $ROOT/bench.py --no-cpu-baseline --no-workloads --min-seconds 0 $ARGS > $OUT/stats_$name.log 2>&1
cp "$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_$name.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_${name}_$c -- python3 $ROOT/bench.py --no-cpu-baseline --no-workloads --no-graph --min-seconds 0 $ARGS --steps 5 --warmup 2 > $OUT/pmc_${name}_$c.log 2>&1
  cp "$(find /tmp/prof_${name}_$c -name '*counter_collection.csv' | head -1)" $OUT/${name}_${c}_counter_collection.csv
done
python3 $ROOT/tools/pmc_summary.py $OUT/${name}_FETCH_SIZE_counter_collection.csv $OUT/${name}_WRITE_SIZE_counter_collection.csv $OUT/pmc_$name.json 655360 32 r06_b "$ARGS" | head -5
tail -1 $OUT/bench_$name.json | cut -c1-200
