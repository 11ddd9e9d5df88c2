#!/bin/bash
# random vs all-zero operands (BENCH_ZERO=1: nothing toggles, the clock stays up) on the same launches: how far a kernel is from its
# power budget.   bash profiles/ab_zero.sh "<bench_one spec>" ...
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for spec in "$@"; do
    for z in "" 1; do
      printf "ZERO=%-1s  " "$z"
      env BENCH_ZERO=$z timeout 120 python profiles/bench_one.py $spec 50 2>&1 | tail -1
    done
  done
done
