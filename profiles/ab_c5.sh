#!/bin/bash
# A/B of library builds on the C5 step inside ONE gpurun call: bash profiles/ab_c5.sh libA.so libB.so ... (files under vnet_tensorflow_amd/)
for rep in 1 2 3; do
  for lib in "$@"; do
    printf "%-24s " "$lib"
    VNET_HIP_LIB=$PWD/vnet_tensorflow_amd/$lib python profiles/step_only.py 60 bf16 4 5 | tail -1
  done
done
