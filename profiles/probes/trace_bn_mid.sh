#!/bin/bash
# NOTE (round 5): the one-launch batch-norm backward this script A/Bs was measured and NOT kept; the kernel and its VNET_BN_MID switch
# exist only in commit b1cb486's parent experiment (see profiles/r04_bn_mid.txt and DESIGN 4.4) -- at HEAD both legs run the same code.
# per-grid durations of the batch-norm backward kernels in the C5 step (VNET_BN_MID=1 / 0) under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 0; do
  mkdir -p gpurun_out/bnmid$v
  VNET_BN_MID=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bnmid$v -o t -- python profiles/step_only.py 20 bf16 4 5 > gpurun_out/bnmid$v/log.txt 2>&1
  python - $v <<'PY'
import csv, sys, glob, collections
v = sys.argv[1]
f = glob.glob("gpurun_out/bnmid%s/**/t_kernel_trace.csv" % v, recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "bn_" in n or "sum_finalize" in n:
        acc[(n, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== VNET_BN_MID=%s" % v)
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    d = acc[k]
    print("%-52s grid %8s  n/step %5.1f  avg %7.1f us  total/step %7.1f us" % (k[0][:52], k[1], len(d) / 24.0, sum(d) / len(d), sum(d) / 24.0))
PY
  rm -rf gpurun_out/bnmid$v
done
