#!/bin/bash
# SQ counter passes (kernel-trace + pmc only) over a short eager bench run; prints per-kernel averages for the edge kernels.
#   bash tools/pmc_sq.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_CYCLES"
n=0
for P in "$P1" "$P2" "$P3"; do
  n=$((n+1))
  rm -rf /tmp/pmc_$n
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_$n -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-workloads --no-graph --min-seconds 0 "$@" > $OUT/pmc_$n.log 2>&1
  cp "$(find /tmp/pmc_$n -name '*counter_collection.csv' | head -1)" $OUT/pmc_$n.csv
done
python3 - $OUT <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for n in (1, 2, 3):
    for r in csv.DictReader(open(f"{out}/pmc_{n}.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "edge_" in k or "knn_" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:32s} {sum(v)/len(v):16.1f}   (n={len(v)})")
PY
