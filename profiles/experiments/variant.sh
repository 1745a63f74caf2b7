#!/bin/bash
# Build a variant of ONE source file with extra -D flags and link it with the current objects into ab/<name>.so
# (A/B runs select it through CVAE_LIB):  bash profiles/experiments/variant.sh <name> <file.hip> [-DX=..]...
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
cs=$root/critic-vae_amd/csrc
make -s -C $cs -j8 > /dev/null
extra=""; [ "$src" = msssim.hip ] && extra="-fno-slp-vectorize"
mkdir -p $root/ab /tmp/variant
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed $extra "$@" -c $cs/$src -o /tmp/variant/$name.o 2>&1 | grep -E "error" || true
objs=$(ls $cs/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ab/$name.so $objs /tmp/variant/$name.o 2>&1 | grep -E "error" | head -3 || true
ls -la $root/ab/$name.so | awk '{print $5, $9}'
