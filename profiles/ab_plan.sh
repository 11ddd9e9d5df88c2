#!/bin/bash
# sweeps of the split planners (experimental build libvnet_env.so, -DVNET_PLAN_ENV) on the deep-level shapes
export VNET_HIP_LIB=$PWD/vnet_tensorflow_amd/libvnet_env.so
for cfg in "256 256" "0 256" "64 256" "1024 256" "256 64" "256 32" "256 16" "0 32" "1024 64"; do
  set -- $cfg
  export VNET_BF16_HALF_MAX=$1 VNET_BF16_NSB_MIN=$2
  echo "== bf16 conv: half bricks up to $1 wide bricks, two cout blocks per workgroup from $2 workgroups"
  for shp in "16 128 128" "16 256 128" "32 64 64" "32 128 64" "64 32 32" "64 64 32"; do
    python profiles/bench_one.py conv bf16 $shp 50 2>&1 | tail -1
  done
done
