#!/bin/bash
# A/B of a library OPTION on the C5 step (bf16 storage, 4 modalities, 5 classes), interleaved:  bash profiles/ab_opt_step.sh NAME v0 v1
cd "$GRAFT_REPO_ROOT"
NAME=$1; V0=$2; V1=$3
for rep in 1 2 3; do
  for v in $V0 $V1; do
    printf "%s=%s  C5 step " $NAME $v
    env VNET_$NAME=$v timeout 300 python bench.py --compute bf16 --channels 4 --classes 5 --no-cpu-baseline --no-sustained --no-c5 --no-c2 --no-x3 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_launch_ms'])"
  done
done
