#!/bin/bash
# interleaved A/B of one environment switch on the C5 step inside ONE gpurun call:  bash profiles/ab_env.sh VAR A B [mode cin K]
VAR=$1; A=$2; B=$3; MODE=${4:-bf16}; CIN=${5:-4}; K=${6:-5}
for rep in 1 2 3; do
  for v in $A $B; do
    printf "%s=%-4s " $VAR $v
    env $VAR=$v python profiles/step_only.py 60 $MODE $CIN $K | tail -1
  done
done
