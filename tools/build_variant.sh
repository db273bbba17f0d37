#!/bin/bash
# Another build of the HIP library for A/B measurements on the GPU box:
#   bash tools/build_variant.sh NAME [extra hipcc flags...]   ->  tools/variants/libpfrl_amd_NAME.so
#   PFRL_AMD_LIB=tools/variants/libpfrl_amd_NAME.so python bench.py --allow-lib-override ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p /tmp/pfrl_var_$NAME tools/variants
for f in pfrl_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
      -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form "$@" -c $f -o /tmp/pfrl_var_$NAME/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/pfrl_var_$NAME/*.o -o tools/variants/libpfrl_amd_$NAME.so
ls -la tools/variants/libpfrl_amd_$NAME.so
