#!/bin/bash
# copy the judged summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked)
#   bash tools/collect_profiles.sh r05_a
TAG=$1; SRC=gpurun_out/$TAG; DST=profiles
cp $SRC/bench.json $DST/${TAG}_bench.json
for w in b1 b16 b16_irreps c4shard c5 c1infer c1infer_b16 irrepgnn irrepgnn_b8; do
  [ -f $SRC/bench_$w.json ] || continue
  cp $SRC/bench_$w.json $DST/${TAG}_bench_$w.json
  cp $SRC/kernel_stats_$w.csv $DST/${TAG}_kernel_stats_$w.csv
  cp $SRC/pmc_$w.json $DST/${TAG}_pmc_$w.json
done
cp $SRC/in_step.json $DST/${TAG}_in_step.json
mkdir -p $DST/${TAG}_pmc && cp $SRC/*_counter_collection.csv $DST/${TAG}_pmc/
cp $SRC/step_sequence.txt $DST/${TAG}_step_sequence.txt
[ -f $SRC/step_sequence_build_ahead.txt ] && cp $SRC/step_sequence_build_ahead.txt $DST/${TAG}_step_sequence_build_ahead.txt
[ -f $SRC/bench_full.json ] && cp $SRC/bench_full.json $DST/${TAG}_bench_full.json
for f in 64 48 16; do [ -f $SRC/kernel_stats_fp_f$f.csv ] && cp $SRC/kernel_stats_fp_f$f.csv $DST/${TAG}_kernel_stats_fp_f$f.csv; done
ls $DST | grep $TAG
