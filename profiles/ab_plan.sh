#!/bin/bash
# sweeps of the split planners (experimental build libvnet_env.so, -DVNET_PLAN_ENV) on the deep-level shapes
export VNET_HIP_LIB=$PWD/vnet_tensorflow_amd/libvnet_env.so
for cfg in "512 256" "256 256" "1024 256" "512 0" "256 0" "1024 0" "2048 0"; do
  set -- $cfg
  export VNET_F32_SPLIT_TARGET=$1 VNET_F32_NZ_MIN=$2
  echo "== fp32 conv target $1 nzmin $2"
  for shp in "8 256 256" "16 128 128" "16 256 128" "32 64 64"; do
    python profiles/bench_one.py conv fp32 $shp 30 2>&1 | tail -1
  done
done
for t in 256 512 1024 128; do
  export VNET_WGRAD_TARGET=$t
  echo "== wgrad target $t"
  for m in fp32 bf16; do
  for shp in "8 256 256" "16 128 128" "32 64 64" "64 32 32"; do
    python profiles/bench_one.py wgrad $m $shp 30 2>&1 | tail -1
  done
  done
done
