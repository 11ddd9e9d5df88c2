#!/bin/bash
# launches, summed kernel time and idle gaps of one graph-replayed training step (rocprofv3 kernel trace of profiles/step_only.py)
#   bash profiles/step_gaps.sh [fp32 1 2 | bf16 4 5]
MODE=${1:-fp32}; CIN=${2:-1}; K=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gaps
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -o t -- python profiles/step_only.py 128 $MODE $CIN $K > gpurun_out/gaps/log.txt 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/gaps/t_kernel_trace.csv')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# steady state: the last 60 % of the trace
ev = ev[int(len(ev) * 0.4):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
ksum = sum(e - s for s, e, _ in ev)
short = [(e - s) for s, e, _ in ev if e - s < 10000]
print("launches %d  wall %.2f ms  GPU busy (union) %.2f ms = %.1f %%  idle %.2f ms  summed kernel time %.2f ms" % (len(ev), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, ksum / 1e6))
print("kernels under 10 us: %d launches, %.2f ms summed (%.1f %% of wall)" % (len(short), sum(short) / 1e6, 100.0 * sum(short) / wall))
d = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    d[n][0] += 1; d[n][1] += e - s
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-72s n=%5d  %7.2f %%  avg %7.1f us" % (n, c, 100.0 * t / wall, t / c / 1e3))
PY
rm -f gpurun_out/gaps/t_kernel_trace.csv
