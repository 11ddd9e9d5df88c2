#!/bin/bash
# Samples rocm-smi power / clocks while the bench runs (read-only queries; nothing is changed on the box).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pc
( timeout 120 python bench.py --no-cpu-baseline --steps 1500 --warmup 10 $BENCH_ARGS > gpurun_out/pc/bench.json 2>/dev/null ) &
BP=$!
sleep 14
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)|Performance" | tr '\n' ';' | sed 's/  */ /g'
  echo
  sleep 2
done
wait $BP
cut -c1-200 gpurun_out/pc/bench.json
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
