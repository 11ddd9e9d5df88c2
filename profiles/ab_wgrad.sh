#!/bin/bash
# interleaved A/B of the row-reuse filter gradient (VNET_WGRAD_RR=2) against the generic kernel (=0) on the C5 layer shapes
cd "$GRAFT_REPO_ROOT"
for rep in 1; do
for shp in "128 16 16" "128 32 16" "128 8 16" "64 32 32" "64 64 32" "32 64 64" "32 128 64"; do
  for rr in 0 1; do
    echo -n "rr=$rr  "; VNET_WGRAD_RR=$rr python profiles/bench_one.py wgrad bf16 $shp 30 2>&1 | tail -1
  done
done
done
