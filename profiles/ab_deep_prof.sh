#!/bin/bash
# In-step A/B of the deep-level kernel under rocprofv3 (same box, back to back): per-kernel time of the C5 step with
# VNET_BF16_DEEP=0 (generic kernels) and =1; prints the 5^3 forward / backward-data kernels and their split-K reduces.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 1 0 1; do
  mkdir -p gpurun_out/abdeep$d
  VNET_BF16_DEEP=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abdeep$d -o c5 -- python profiles/step_only.py 100 bf16 4 5 > gpurun_out/abdeep$d/c5.log 2>&1
  rm -f gpurun_out/abdeep$d/*_kernel_trace.csv
  python - $d <<'PY'
import csv, sys
d = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/abdeep%s/c5_kernel_stats.csv" % d)))
n = 104.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / n
sel = [r for r in rows if ("conv5_bf16_kernel" in r["Name"] or "deep" in r["Name"] or "splitk_reduce_b16" in r["Name"])]
print("DEEP=%s  step %.3f ms of kernels;" % (d, tot), "  ".join("%s x%.0f %.1fus" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3) for r in sel),
      " => %.3f ms/step" % (sum(float(r["TotalDurationNs"]) for r in sel) / 1e6 / n))
PY
done
