#!/bin/bash
# Debug build of the HIP library with in-kernel phase clocks (tools/per_dbg.py, tools/qnet_phase.py):
#   bash tools/build_dbg.sh   ->  tools/libpfrl_amd_dbg.so
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/pfrl_dbg
for f in pfrl_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form \
      -DPFRL_TREE_DEBUG -DPFRL_QNET_DEBUG -c $f -o /tmp/pfrl_dbg/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pfrl_dbg/*.o -o tools/libpfrl_amd_dbg.so
ls -la tools/libpfrl_amd_dbg.so
