#!/bin/bash
# Evidence for "the edge kernels are issue-bound" (DESIGN section 4), collected in one GPU call:
#   SQ counter passes (tools/pmc_sq.sh: kernel-trace + pmc only) over the edge kernels at one pair, 16 pairs and the KITTI shard,
#   and the co-execution probe tools/dev/coexec.hip (fp32 MFMA vs VALU on one SIMD).
#   gpurun --timeout 1500 -- 'bash tools/sq_evidence.sh r03'   ->  gpurun_out/<tag>_sq/{b1,b16,c4}.txt, coexec.txt
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_sq; mkdir -p $OUT
bash $ROOT/tools/pmc_sq.sh ${TAG}_sq_b1 > $OUT/b1.txt 2>&1
bash $ROOT/tools/pmc_sq.sh ${TAG}_sq_b16 --workload c3 > $OUT/b16.txt 2>&1
bash $ROOT/tools/pmc_sq.sh ${TAG}_sq_c4 --workload c4 > $OUT/c4.txt 2>&1
bash $ROOT/tools/pmc_sq.sh ${TAG}_sq_c5 --workload c5 > $OUT/c5.txt 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $ROOT/tools/dev/coexec.hip -o /tmp/coexec && /tmp/coexec > $OUT/coexec.txt 2>&1
tail -n 40 $OUT/b16.txt; cat $OUT/coexec.txt
