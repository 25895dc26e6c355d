#!/bin/bash
# Builds loss_kernels.hip for (n,p,K) = (4,2,3) alone under different compiler options and
# links each into a copy of the library under tools/exp/_build/ (KVAE_LIB_PATH selects one).
set -e
cd "$(dirname "$0")/../../kalman_vae_amd/csrc"
OUT=../../tools/exp/_build
mkdir -p $OUT
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function"
OTHERS=$(ls *.o | grep -v '^loss_kernels.o$')
build() {
  name=$1; shift
  /opt/rocm/bin/hipcc $F "-DKVAE_DIM_LIST(X)=X(4,2,3)" "$@" -c -o $OUT/loss_$name.o loss_kernels.hip
  /opt/rocm/bin/hipcc $F -shared -o $OUT/libkvae_$name.so $OTHERS $OUT/loss_$name.o
}
build shipped -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-ilp &
build vgprform -mllvm -amdgpu-mfma-vgpr-form &
build plain &
build maxilp -mllvm -amdgpu-sched-strategy=max-ilp &
build maxocc -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=max-occupancy &
wait
ls $OUT/*.so
