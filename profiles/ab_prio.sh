#!/bin/bash
# A/B of the bf16 priority-alternation experiment builds (-DVNET_BF16_PRIO=n, profiles/build_stamps.sh) against the shipped library,
# interleaved inside ONE gpurun call: the 128^3 family launches and the C5 step.   bash profiles/ab_prio.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT"
LIBS="$PWD/vnet_tensorflow_amd/libvnet_hip.so"
for l in "$@"; do LIBS="$LIBS $PWD/profiles/probes/$l"; done
for rep in 1 2 3; do
  for lib in $LIBS; do
    n=$(basename $lib)
    for spec in "conv bf16 128 32 16" "conv bf16 128 16 16" "wgrad bf16 128 32 16"; do
      printf "%-24s %-22s " "$n" "$spec"
      VNET_HIP_LIB=$lib timeout 120 python profiles/bench_one.py $spec 50 2>&1 | tail -1
    done
    printf "%-24s C5 step " "$n"
    VNET_HIP_LIB=$lib timeout 300 python bench.py --compute bf16 --channels 4 --classes 5 --no-cpu-baseline --no-sustained --no-c5 --no-c2 --no-x3 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_launch_ms'])"
  done
done
