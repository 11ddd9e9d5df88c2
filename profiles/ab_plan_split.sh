#!/bin/bash
# split-K of the bf16-storage conv at exactly 256 workgroups (the 32^3 level); needs the -DVNET_PLAN_ENV build:
#   DEFS=-DVNET_PLAN_ENV OUT=libvnet_hip_env.so bash profiles/build_stamps.sh
export VNET_HIP_LIB=$PWD/profiles/probes/libvnet_hip_env.so
for cfg in "255 512" "256 512" "256 1024" "512 1024" "512 2048"; do
  set -- $cfg
  export VNET_BF16_SPLIT_NWG_MAX=$1 VNET_BF16_SPLIT_TARGET=$2
  echo "== split when nwg <= $1, target $2"
  for shp in "32 64 64" "32 128 64" "32 64 128" "16 128 128" "64 32 32"; do
    python profiles/bench_one.py conv bf16 $shp 100 2>&1 | tail -1
  done
done
