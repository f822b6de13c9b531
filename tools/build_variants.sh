#!/bin/bash
# Variant builds of the conv tile loop into gauspcc_amd/variants/ (git-ignored; they travel to the GPU box), one
# libgauspcc_<name>.so each.  A spec is  name[:GENERATOR_ENV=VALUE[,ENV=VALUE...]][:-DFLAG[,-DFLAG...]]
#   a7:CONV_ASM_EXP=7          ablation: no LDS sums, no weight loads, no gathers (results are WRONG on purpose)
#   rb128:CONV_ASM_ROWB=128    LDS row pitch of 128 bytes (the bank-conflicted layout of rounds 1-2)
#   timing::-DCONV_TIMING      shader-clock stamps per phase
#   noepi::-DCONV_NO_EPILOGUE  ablation: no copy-out (results are WRONG on purpose)
# (the other objects come from the regular build: run `make -C gauspcc_amd/csrc` first; pair-loop ablations: CONV_ASM_EXP2=.. tools/gen_conv_loop2.py, then
#  rebuild network.o by hand and regenerate the include)
# Time them with  GAUSPCC_LIB=gauspcc_amd/variants/libgauspcc_<name>.so python tools/enc_only.py
# Usage: tools/build_variants.sh spec [spec ...]
set -e
cd "$(dirname "$0")/../gauspcc_amd/csrc"
mkdir -p ../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wno-unused-result"
names=()
for spec in "$@"; do
  IFS=: read -r name genv defs <<< "$spec"
  names+=("$name")
  (
    for kv in ${genv//,/ }; do export "$kv"; done
    CONV_ASM_OUT=$PWD/../variants/conv_loop_$name.inc python3 ../../tools/gen_conv_loop.py > /dev/null
    /opt/rocm/bin/hipcc $FLAGS ${defs//,/ } -DCONV_LOOP_INC="\"../variants/conv_loop_$name.inc\"" -c network.hip -o ../variants/network_$name.o
    /opt/rocm/bin/hipcc $FLAGS ${defs//,/ } -DCONV_LOOP_INC="\"../variants/conv_loop_$name.inc\"" -c tiles.hip -o ../variants/tiles_$name.o
  ) &
done
wait
for name in "${names[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libgauspcc_$name.so primitives.o octree.o ../variants/tiles_$name.o ../variants/network_$name.o network_any.o rangecoder.o hostcoder.o codec.o forest.o codec_batch.o fused.o api.o attributes.o rasterizer.o neural_gaussians.o
done
ls -la ../variants/*.so
