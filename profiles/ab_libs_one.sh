#!/bin/bash
# several library builds (shipped + profiles/probes/<name>...) on bench_one specs, interleaved:  bash profiles/ab_libs_one.sh "lib1.so lib2.so" "<spec>" ...
cd "$GRAFT_REPO_ROOT"
LIBS="$PWD/vnet_tensorflow_amd/libvnet_hip.so"
for l in $1; do LIBS="$LIBS $PWD/profiles/probes/$l"; done
shift
for rep in 1 2; do
  for lib in $LIBS; do
    for spec in "$@"; do
      printf "%-28s " "$(basename $lib)"
      VNET_HIP_LIB=$lib timeout 120 python profiles/bench_one.py $spec 50 2>&1 | tail -1
    done
  done
done
