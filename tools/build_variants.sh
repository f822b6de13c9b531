#!/bin/bash
# Ablation builds of the conv tile loop: tools/gen_conv_loop.py with CONV_ASM_EXP=<mask> (1 no LDS sums, 2 no weight loads,
# 4 no gathers) into gauspcc_amd/variants/, one libgauspcc_a<mask>.so each.  The results are WRONG on purpose; time them with
#   GAUSPCC_LIB=gauspcc_amd/variants/libgauspcc_a7.so python tools/enc_only.py
# Usage: tools/build_variants.sh 1 2 4 6 7
set -e
cd "$(dirname "$0")/../gauspcc_amd/csrc"
mkdir -p ../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wno-unused-result"
for v in "$@"; do
  CONV_ASM_EXP=$v CONV_ASM_OUT=$PWD/../variants/conv_loop_a$v.inc python3 ../../tools/gen_conv_loop.py > /dev/null
  /opt/rocm/bin/hipcc $FLAGS -DCONV_LOOP_INC="\"../variants/conv_loop_a$v.inc\"" -c network.hip -o ../variants/network_a$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libgauspcc_a$v.so primitives.o octree.o tiles.o ../variants/network_a$v.o rangecoder.o codec.o api.o attributes.o rasterizer.o neural_gaussians.o
done
ls -la ../variants/*.so
