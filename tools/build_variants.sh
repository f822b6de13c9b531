#!/bin/bash
# Build experiment variants of libgauspcc.so (network.hip compiled with -DCONV_EXP=<mask>) into gauspcc_amd/variants/.
# Usage: tools/build_variants.sh 1 2 4 7 ...   then   GAUSPCC_LIB=gauspcc_amd/variants/libgauspcc_exp1.so python tools/enc_only.py
set -e
cd "$(dirname "$0")/../gauspcc_amd/csrc"
mkdir -p ../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wno-unused-result"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DCONV_EXP=$v $EXTRA -c network.hip -o ../variants/network_exp$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libgauspcc_exp$v.so primitives.o octree.o ../variants/network_exp$v.o rangecoder.o codec.o api.o attributes.o rasterizer.o
done
ls -la ../variants/*.so
