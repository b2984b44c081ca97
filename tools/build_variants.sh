#!/bin/bash
# Build A/B variants of one kernel source: variants/libsage_gfx950_<tag>.so (git-ignored by *.so).
# usage: [VARIANT_SRC=sage_prepass.hip] tools/build_variants.sh tag1:"-DSAGE_X=1 ..." tag2:"..."
# VARIANT_SRC (default sage_attn.hip) is compiled with the flags; every other object comes from the regular build.
set -e
cd "$(dirname "$0")/.."
SRC="${VARIANT_SRC:-sage_attn.hip}"
STEM="${SRC%.hip}"
mkdir -p variants
make -C sageattention_amd/csrc -j8 -s
OTHERS=$(ls sageattention_amd/csrc/*.o | grep -v "/$STEM.o")
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden $flags \
      -c sageattention_amd/csrc/$SRC -o variants/${STEM}_$tag.o \
      -Rpass-analysis=kernel-resource-usage 2> variants/${STEM}_$tag.res && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsage_gfx950_$tag.so variants/${STEM}_$tag.o $OTHERS && \
    echo "built $tag: $(grep -A12 'ILi128ELb1ELb1ELb1ELb1E\|prepass_kv_kernelILi128ELi1E' variants/${STEM}_$tag.res | grep -E 'VGPRs:|Occupancy|Spill' | sed 's/.*remark: [^ ]* *//' | tr '\n' ' ')" ) &
done
wait
