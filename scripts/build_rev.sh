#!/bin/bash
# Build the library of another git revision next to the product one (same-box A/B runs: WNV_LIB=<out> python bench.py ...).
#   scripts/build_rev.sh <git-rev> <out.so> [extra hipcc flags]
set -e
rev=$1; out=$(realpath -m "$2"); shift 2
tmp=$(mktemp -d)
git -C "$(dirname "$0")/.." archive "$rev" wavenet_vocoder_amd/csrc include | tar -x -C "$tmp"
srcs=""
for f in wnv_host.cpp wnv_layers.cpp wnv_generic.hip wnv_upsample.hip wnv_ring.hip wnv_wide.hip wnv_post.hip wnv_forward.hip wnv_mel.hip wnv_ubench.hip; do [ -f "$tmp/wavenet_vocoder_amd/csrc/$f" ] && srcs="$srcs $tmp/wavenet_vocoder_amd/csrc/$f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function "$@" -o "$out" -x hip $srcs
rm -rf "$tmp"
echo "$out"
