#!/bin/bash
# Build A/B variants of the library: variants/libsage_gfx950_<tag>.so (git-ignored by *.so).
# usage: [VARIANT_SRC="sage_prepass.hip"] tools/build_variants.sh tag1:"-DSAGE_X=1 ..." tag2:"..."
# VARIANT_SRC (default: the attention units sage_attn_d*.hip + sage_attn.hip) is compiled with the flags; every other object comes from the
# regular build.
set -e
cd "$(dirname "$0")/.."
CS=sageattention_amd/csrc
SRCS="${VARIANT_SRC:-$(cd $CS && ls sage_attn_d*.hip | tr '\n' ' ') sage_attn.hip}"
mkdir -p variants
make -C $CS -j8 -s
OTHERS=""
for o in $CS/*.o; do
  s="$(basename "${o%.o}").hip"
  case " $SRCS " in *" $s "*) ;; *) OTHERS="$OTHERS $o";; esac
done
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  (
    objs=""
    for s in $SRCS; do
      stem="${s%.hip}"
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden $flags -c $CS/$s -o variants/${stem}_$tag.o \
          -Rpass-analysis=kernel-resource-usage 2> variants/${stem}_$tag.res &
      objs="$objs variants/${stem}_$tag.o"
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsage_gfx950_$tag.so $objs $OTHERS && \
    echo "built $tag: scratch $(cat variants/*_$tag.res | grep -c 'ScratchSize.*: [1-9]') spills $(cat variants/*_$tag.res | grep -c 'VGPRs Spill: [1-9]') kernels $(cat variants/*_$tag.res | grep -c 'Function Name')"
  ) &
done
wait
