#!/bin/bash
# A/B of library builds inside ONE gpurun call:  bash profiles/ab_lib.sh libA.so libB.so ...   (files under vnet_tensorflow_amd/)
for rep in 1 2; do
  for lib in "$@"; do
    printf "%-22s " "$lib"
    VNET_HIP_LIB=$PWD/vnet_tensorflow_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['c5_bf16']['ms_per_step'], d['c5_bf16']['roofline']['frac'], d['final_loss'])"
  done
done
