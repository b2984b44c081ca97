#!/bin/bash
# Build A/B variants of the attention kernel: libsage_gfx950_<tag>.so in gpurun_variants/ (git-ignored by *.so).
# usage: tools/build_variants.sh tag1:"-DSAGE_X=1 ..." tag2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
make -C sageattention_amd/csrc -j8 -s
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden $flags \
      -c sageattention_amd/csrc/sage_attn.hip -o variants/attn_$tag.o \
      -Rpass-analysis=kernel-resource-usage 2> variants/attn_$tag.res && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsage_gfx950_$tag.so variants/attn_$tag.o \
      sageattention_amd/csrc/sage_cabi.o sageattention_amd/csrc/sage_quant.o sageattention_amd/csrc/sage_prep_v.o sageattention_amd/csrc/sage_stats.o sageattention_amd/csrc/sage_merge.o && \
    echo "built $tag: $(grep -A12 'ILi128ELb1ELb1ELb1ELb1E' variants/attn_$tag.res | grep -E 'VGPRs:|Occupancy|Spill' | sed 's/.*remark: [^ ]* *//' | tr '\n' ' ')" ) &
done
wait
