#!/bin/bash
# Development tool: builds the kernel library of another git revision (or of the working tree with extra -D flags) as
# synthesis-in-style_amd/lib/libsis_hip_<tag>.so, for same-box A/B timing (tools/bench_layers.py picks it with SIS_HIP_LIB).
#   tools/build_variant.sh <git-rev|WORK> <tag> [extra hipcc flags]
set -e
rev=$1; tag=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
# (the sources include "../../include/sis_hip.h": keep the tree's depth)
if [ "$rev" = WORK ]; then mkdir -p "$tmp/synthesis-in-style_amd" "$tmp/include"; cp -r "$root/synthesis-in-style_amd/csrc" "$tmp/synthesis-in-style_amd/csrc"; cp "$root/include/sis_hip.h" "$tmp/include/";
else git -C "$root" archive "$rev" synthesis-in-style_amd/csrc include | tar -x -C "$tmp"; fi
cd "$tmp/synthesis-in-style_amd/csrc"; rm -rf _obj _obj_trace
objs=""
for f in *.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I"$tmp/include" "$@" -c "$f" -o "${f%.hip}.o" & objs="$objs ${f%.hip}.o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/synthesis-in-style_amd/lib/libsis_hip_$tag.so" $objs
rm -rf "$tmp"; echo "built lib/libsis_hip_$tag.so from $rev"
