#!/bin/bash
# Builds loss_row_kernels.hip for (n,p,K) = (4,2,3) alone in variants that stop after phase
# F / S / SA of a sequence (KR_STOP_AFTER) and links each into a copy of the library under
# tools/exp/_build/ (KVAE_LIB_PATH selects one): the differences of their run times are the
# run times of the four phases.  Extra arguments are passed to every compile.
set -e
cd "$(dirname "$0")/../../kalman_vae_amd/csrc"
OUT=../../tools/exp/_build
mkdir -p $OUT
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -fno-slp-vectorize"
OTHERS=$(ls *.o | grep -v '^loss_row_kernels.o$')
build() {
  name=$1; shift
  /opt/rocm/bin/hipcc $F "-DKVAE_ROW_DIM_LIST(X)=X(4,2,3)" "$@" -c -o $OUT/row_$name.o loss_row_kernels.hip
  /opt/rocm/bin/hipcc $F -shared -o $OUT/libkvae_row_$name.so $OTHERS $OUT/row_$name.o
}
for n in 1 2 3 4; do build stop$n -DKR_STOP_AFTER=$n "$@" & done
wait
ls $OUT/libkvae_row_*.so
