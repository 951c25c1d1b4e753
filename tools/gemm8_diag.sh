#!/bin/bash
# Diagnostic builds of the deep-pipelined GEMM (csrc/gemm8.hip MMHIP_DIAG_*: WRONG results -- each removes one ingredient of the K loop) and their
# timings on the step's shapes beside the product build, same process-by-process on one box.  Usage (GPU box): bash tools/gemm8_diag.sh > out.txt
set -e
P=socialmedia-textimage-classification-auxlosses_amd
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJS=$(ls $P/build/*.o | grep -v gemm8)
for d in ${DIAGS:-NOBARRIER NOWAITVM NODMA NOLDS NOSTORE}; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMMHIP_DIAG_$d -c $P/csrc/gemm8.hip -o /tmp/gemm8_$d.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm8_$d.o -o /tmp/libmmhip_$d.so
done
export SHAPES="${SHAPES:-txt qkv,vit qkv,vit fc1,square 8192}" VARIANTS="${VARIANTS:-15,18}" ROUNDS=${ROUNDS:-9}
echo "== product build"; python tools/gemm8_bench.py
for d in ${DIAGS:-NOBARRIER NOWAITVM NODMA NOLDS NOSTORE}; do echo "== $d"; MMHIP_LIB_PATH=/tmp/libmmhip_$d.so python tools/gemm8_bench.py; done
