#!/bin/bash
# Build a complete libsage_gfx950_<tag>.so with extra flags on EVERY source (for constants shared through headers).
# usage: tools/build_full_variant.sh tag "-DSAGE_STATS_SLAB=256 ..."
set -e
cd "$(dirname "$0")/.."
tag="$1"; flags="$2"
mkdir -p variants/full_$tag
for f in sageattention_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden $flags -c $f -o variants/full_$tag/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libsage_gfx950_$tag.so variants/full_$tag/*.o
echo "built variants/libsage_gfx950_$tag.so"
