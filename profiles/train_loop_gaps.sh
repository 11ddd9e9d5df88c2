#!/bin/bash
# where the product loop loses time against the replayed step alone: kernel trace of profiles/train_loop_bench.py,
# idle time between consecutive steps (a step = the kernels between two adam_kernel launches)
MODE=${1:-bf16}; CIN=${2:-4}; K=${3:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tlg
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlg -o t -- python profiles/train_loop_bench.py 128 $MODE $CIN $K > gpurun_out/tlg/log.txt 2>&1
tail -1 gpurun_out/tlg/log.txt
python - <<'PY'
import csv
k = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open('gpurun_out/tlg/t_kernel_trace.csv')))
ends = [i for i, e in enumerate(k) if 'adam_kernel' in e[2]]
# steady state: the last 40 steps
ends = ends[-41:]
tot = gap_sum = 0
biggest = []
for a, b in zip(ends[:-1], ends[1:]):
    ev = k[a + 1:b + 1]
    t0, t1 = k[a][1], ev[-1][1]
    busy = 0; cs, ce = ev[0][0], ev[0][1]
    gaps = [(ev[0][0] - t0, 'step start: after adam -> ' + ev[0][2][:40])]
    for s, e, n in ev[1:]:
        if s > ce:
            gaps.append((s - ce, n[:50])); busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tot += t1 - t0; gap_sum += (t1 - t0) - busy
    biggest += gaps
print("steps %d  mean step %.3f ms  idle per step %.3f ms" % (len(ends) - 1, tot / (len(ends) - 1) / 1e6, gap_sum / (len(ends) - 1) / 1e6))
import collections
d = collections.defaultdict(lambda: [0, 0])
for g, n in biggest:
    if g > 2000:
        d[n][0] += 1; d[n][1] += g
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  idle before %-52s n=%4d  %.3f ms per step" % (n, c, t / (len(ends) - 1) / 1e6))
blit = [(s, e) for s, e, n in k[ends[0]:ends[-1]] if 'copyBuffer' in n]
print("blit kernels (__amd_rocclr_copyBuffer): %.1f per step, %.3f ms per step summed" % (len(blit) / (len(ends) - 1), sum(e - s for s, e in blit) / (len(ends) - 1) / 1e6))
PY
rm -f gpurun_out/tlg/t_kernel_trace.csv
