#!/bin/bash
# A/B of environment switches inside ONE gpurun call (box-to-box variation is ~0.5 %):  bash profiles/ab.sh "A=1" "B=2 C=3" ...
# each argument is one variant (space-separated VAR=value list); every variant is run twice, interleaved.
for rep in 1 2; do
  for v in "$@"; do
    printf "%-60s " "$v"
    env $v timeout 200 python bench.py --no-cpu-baseline --steps 20 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])"
  done
done
