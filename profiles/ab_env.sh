#!/bin/bash
# interleaved A/B of whole training steps (graph replay, profiles/step_only.py) between two settings of ONE environment variable:
#   bash profiles/ab_env.sh <VAR> "<value> <value> ..." <compute> [channels classes] [steps]
cd "$GRAFT_REPO_ROOT"
VAR=$1; VALS=$2; COMPUTE=${3:-fp32_split3}; CH=${4:-1}; K=${5:-2}; STEPS=${6:-60}
for rep in 1 2 3; do
  for v in $VALS; do
    printf "%-24s %-12s " "$VAR=$v" "$COMPUTE"
    env $VAR=$v timeout 300 python profiles/step_only.py $STEPS $COMPUTE $CH $K 2>&1 | tail -1
  done
done
