#!/bin/bash
# A/B of the deep-level bf16 kernel (csrc/conv_deep.h) on the C5 step inside ONE gpurun call, interleaved: VNET_BF16_DEEP=0 / 1
for rep in 1 2 3; do
  for d in 0 1; do
    printf "VNET_BF16_DEEP=%s  " "$d"
    VNET_BF16_DEEP=$d python profiles/step_only.py 100 bf16 4 5 | tail -1
  done
done
