#!/bin/bash
# compile one .hip of sageattention_amd/csrc for gfx950 with -save-temps into /tmp and print per-kernel register statistics
# usage: tools/asm_stats.sh sage_attn64.hip [extra hipcc flags]
set -e
src=$1; shift
out=/tmp/asm_stats/$(basename $src .hip); mkdir -p $out
cd "$(dirname "$0")/../sageattention_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function "$@" -c $src -o $out/x.o -save-temps=obj
S=$(ls $out/*-hip-amdgcn-amd-amdhsa-gfx950.s)
echo "asm: $S"
awk '/^_Z.*:.*@/ {name=$1} /^; NumVgprs:/ {v=$3} /^; NumAgprs:/ {a=$3} /^; ScratchSize:/ {sc=$3} /^; Occupancy:/ {printf "%-90s vgpr %s agpr %s scratch %s occ %s\n", substr(name,1,90), v, a, sc, $3}' $S
