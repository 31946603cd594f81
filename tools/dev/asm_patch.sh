#!/bin/bash
# Rebuild egnn_bwd.o from a (possibly edited) device assembly listing, then link a variant library: the tool behind the
# ISA-level bisection of profiles/r06_mfma_hazard_probe.txt.
#   tools/dev/asm_patch.sh <device .s> <name> [host compile defines...]   ->  se3_equi_graph_registration_amd/libequigspr_<name>.so
set -e
S=$1; NAME=$2; shift 2
ROOT=$(cd $(dirname $0)/../.. && pwd); C=$ROOT/se3_equi_graph_registration_amd/csrc
LL=/opt/rocm/lib/llvm/bin; W=/tmp/asmx/$NAME; mkdir -p $W
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $S -o $W/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $C/egnn_bwd.hip -o $W/egnn_bwd.o 2>&1 | grep -i "error" || true
OBJS=$(ls $C/*.o | grep -v "egnn_bwd.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $W/egnn_bwd.o -o $C/../libequigspr_$NAME.so
ls -la $C/../libequigspr_$NAME.so
