#!/bin/bash
# A/B of a library OPTION inside ONE gpurun call (interleaved):  bash profiles/ab_opt.sh NAME v0 v1 -- "<bench_one spec>" ...
cd "$GRAFT_REPO_ROOT"
NAME=$1; V0=$2; V1=$3; shift 4
for rep in 1 2 3; do
  for v in $V0 $V1; do
    for spec in "$@"; do
      printf "%s=%s  " $NAME $v
      env VNET_$NAME=$v timeout 120 python profiles/bench_one.py $spec 50 2>&1 | tail -1
    done
  done
done
