#!/bin/bash
# planner sweep on the deep-level bf16-storage convolutions (experiment build -DVNET_PLAN_ENV: profiles/build_stamps.sh with
# EXTRA=-DVNET_PLAN_ENV OUT=libvnet_hip_env.so); prints ms per conv incl. its split-K reduce
export VNET_HIP_LIB=$PWD/profiles/probes/libvnet_hip_env.so
for cfg in "256 512 256 64" "256 256 256 64" "256 1024 256 64" "32 512 256 64" "32 256 256 64" "16 512 256 64" "16 256 256 64" "256 512 64 64" "256 512 16 64" "32 512 64 64"; do
  set -- $cfg
  export VNET_BF16_NSB_MIN=$1 VNET_BF16_SPLIT_TARGET=$2 VNET_BF16_HALF_MAX=$3 VNET_BF16_NZ_MIN=$4
  echo "== nsb_min $1 split_target $2 half_max $3 nz_min $4"
  for shp in "32 64 64" "32 128 64" "16 128 128" "16 256 128" "8 256 256"; do
    python profiles/bench_one.py conv bf16 $shp 100 2>&1 | tail -1
  done
done
