#!/bin/bash
# row-pair bf16 kernel on/off (experimental build libvnet_env.so, -DVNET_PLAN_ENV; VNET_BF16_R32=0 switches it off)
export VNET_HIP_LIB=$PWD/vnet_tensorflow_amd/libvnet_env.so
for rep in 1 2; do
for r in 1 0; do
  export VNET_BF16_R32=$r
  echo "== row-pair kernel $r"
  for shp in "128 16 32" "64 32 32" "64 64 32" "64 32 64" "128 32 32"; do
    python profiles/bench_one.py conv bf16 $shp 40 2>&1 | tail -1
  done
done
done
