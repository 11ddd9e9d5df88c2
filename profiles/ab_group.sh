#!/bin/bash
# A/B of the grouped launch of the deep-level filter gradients (vnet_conv_wgrad_b16_group) on the C5 step inside ONE gpurun call,
# interleaved: VNET_WGRAD_GROUP=0 / 1 (optionally VNET_WGRAD_GROUP_ROUNDS as $1 for the =1 leg); then the same under rocprofv3:
# filter-gradient kernels + slab reduce, ms per step
R=${1:-2}
for rep in 1 2 3; do
  for d in 0 1; do
    printf "VNET_WGRAD_GROUP=%s  " "$d"
    VNET_WGRAD_GROUP=$d VNET_WGRAD_GROUP_ROUNDS=$R python profiles/step_only.py 100 bf16 4 5 | tail -1
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 1; do
  mkdir -p gpurun_out/abgroup$d
  VNET_WGRAD_GROUP=$d VNET_WGRAD_GROUP_ROUNDS=$R rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abgroup$d -o c5 -- python profiles/step_only.py 100 bf16 4 5 > gpurun_out/abgroup$d/c5.log 2>&1
  rm -f gpurun_out/abgroup$d/*_kernel_trace.csv
  python - $d <<'PY'
import csv, sys
d = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/abgroup%s/c5_kernel_stats.csv" % d)))
n = 104.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / n
sel = [r for r in rows if "wgrad" in r["Name"] and "wgrad_kernel<2" not in r["Name"]]
print("GROUP=%s  step %.3f ms of kernels;" % (d, tot), "  ".join("%s x%.0f %.1fus" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3) for r in sel),
      " => %.3f ms/step" % (sum(float(r["TotalDurationNs"]) for r in sel) / 1e6 / n))
PY
done
