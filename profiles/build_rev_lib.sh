#!/bin/bash
# Build libvnet_hip.so of another git revision next to the shipped one, for interleaved A/B runs (VNET_HIP_LIB=...):
#   bash profiles/build_rev_lib.sh <git rev> <tag>   ->  vnet_tensorflow_amd/libvnet_hip_<tag>.so   (git-ignored, travels with gpurun)
set -e
cd "$(dirname "$0")/.."
REV=$1; TAG=$2
TMP=$(mktemp -d)
git archive "$REV" vnet_tensorflow_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/vnet_tensorflow_amd/csrc" -j4 ../libvnet_hip.so >/dev/null 2>&1
cp "$TMP/vnet_tensorflow_amd/libvnet_hip.so" "vnet_tensorflow_amd/libvnet_hip_$TAG.so"
rm -rf "$TMP"
ls -la "vnet_tensorflow_amd/libvnet_hip_$TAG.so"
