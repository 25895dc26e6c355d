#!/bin/bash
# Builds the experiment variants tools/exp_lstm_resident.py times (results of these builds are
# wrong on purpose: they leave out the stores, the activations or the barrier).
cd "$(dirname "$0")/../../kalman_vae_amd/csrc" || exit 1
for v in BASE NOSTORE NOACT NOBARRIER "NOSTORE -DKVAE_EXP_NOACT"; do
  n=$(echo $v | tr -d ' -' | sed 's/DKVAE_EXP_//')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DKVAE_EXP_$v -shared \
      -o ../../tools/exp/liblstm_$n.so lstm_resident.hip
done
