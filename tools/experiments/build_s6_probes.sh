#!/bin/bash
# Experiment builds of libquickprefill.so that differ from the product in ONE define of attn_fwd_kernel_s6 (wrong results by design:
# upper bounds for what removing softmax VALU work could buy).  -> tools/experiments/build/libqp_<tag>.so (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/../../quickvideo_amd/csrc"
make -s
OUT=../../tools/experiments/build; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OTHERS=$(ls *.o | grep -v '^qp_attn_s6.o$')
build() { tag=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c qp_attn_s6.hip -o $OUT/qp_attn_s6_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libqp_$tag.so $OTHERS $OUT/qp_attn_s6_$tag.o -L/opt/rocm/lib -lhipblaslt
  echo built $OUT/libqp_$tag.so; }
build nofma -DQP_S6_NOFMA
build nosum -DQP_S6_NOSUM
build nofma_nosum -DQP_S6_NOFMA -DQP_S6_NOSUM
build dot2sum -DQP_S6_DOT2SUM
