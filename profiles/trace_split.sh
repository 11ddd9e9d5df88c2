#!/bin/bash
# rocprofv3 kernel stats of the 32^3 64->64 bf16-storage conv with and without the 256-workgroup split-K (two workgroups per CU);
# needs the -DVNET_PLAN_ENV build: DEFS=-DVNET_PLAN_ENV OUT=libvnet_hip_env.so bash profiles/build_stamps.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export VNET_HIP_LIB=$PWD/profiles/probes/libvnet_hip_env.so
for m in 255 256; do
  export VNET_BF16_SPLIT_NWG_MAX=$m
  rm -rf gpurun_out/sk; mkdir -p gpurun_out/sk
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sk -o t -- python profiles/bench_one.py conv bf16 32 64 64 100 > gpurun_out/sk/log.txt 2>&1
  echo "== split when nwg <= $m"; tail -1 gpurun_out/sk/log.txt
  python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/sk/t_kernel_stats.csv')):
    if 'conv5_bf16' in r['Name'] or 'splitk' in r['Name']:
        print("  %-90s calls %s avg %.1f us" % (r['Name'].replace('(anonymous namespace)::','')[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
