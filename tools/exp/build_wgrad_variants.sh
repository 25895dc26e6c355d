#!/bin/bash
# Builds kalman_vae_amd/csrc/wgrad_tn.hip with parts compiled out or other ring depths
# into tools/exp/_build/*.so (results of the NOMFMA / NOPUT builds are wrong on purpose).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/exp/_build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -shared"
for v in base:"" nomfma:"-DWG_EXP_NOMFMA" noput:"-DWG_EXP_NOPUT" d3:"-DWG_D=3" noput_nomfma:"-DWG_EXP_NOPUT -DWG_EXP_NOMFMA"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F $flags -o tools/exp/_build/wgrad_$name.so kalman_vae_amd/csrc/wgrad_tn.hip &
done
wait
ls tools/exp/_build
