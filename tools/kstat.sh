#!/bin/bash
# per-kernel average durations of a bench run: kstat.sh <tag> <bench args...>   (EQUIGSPR_LIB selects a library variant)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/kstat_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-workloads --min-seconds 0 "$@" > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print("== $TAG", "$*")
for r in rows[:28]:
    print("  %-46s calls %5s avg %9.1f us  %6s%%" % (r["Name"].replace("void ", "").replace("eq::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
