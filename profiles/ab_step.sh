#!/bin/bash
# interleaved A/B of whole training steps (graph replay, profiles/step_only.py) between library builds:
#   bash profiles/ab_step.sh "<lib.so> <lib.so> ..." <compute> [channels classes] [steps]
cd "$GRAFT_REPO_ROOT"
LIBS=$1; COMPUTE=${2:-fp32_split3}; CH=${3:-1}; K=${4:-2}; STEPS=${5:-60}
for rep in 1 2 3; do
  for lib in $LIBS; do
    printf "%-28s %-12s " "$lib" "$COMPUTE"
    VNET_HIP_LIB=$PWD/vnet_tensorflow_amd/$lib timeout 300 python profiles/step_only.py $STEPS $COMPUTE $CH $K 2>&1 | tail -1
  done
done
