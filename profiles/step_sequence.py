"""The launch sequence of ONE graph-replayed training step from a rocprofv3 kernel trace (profiles/step_gaps.sh leaves it in
gpurun_out/gaps/t_kernel_trace.csv): every launch with its duration and the idle gap in front of it; then the short launches grouped
by (kernel, previous kernel).  python profiles/step_sequence.py [trace.csv] > gpurun_out/r06_step_sequence.txt"""
import collections
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gaps/t_kernel_trace.csv"
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
# one step = from one softmax_dice_fwd to the next; take the second to last one
starts = [i for i, e in enumerate(ev) if short(e[2]).startswith("softmax_dice_fwd")]
a, b = starts[-3], starts[-2]
step = ev[a:b]
wall = step[-1][1] - step[0][0]
print("# one step: %d launches, %.3f ms from the first launch's start to the last one's end" % (len(step), wall / 1e6))
prev_end, prev_name = step[0][0], "-"
tiny = collections.defaultdict(lambda: [0, 0, 0])
gap_total = 0
for s, e, n in step:
    gap = max(0, s - prev_end)
    gap_total += gap
    print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, gap / 1e3, short(n)))
    if e - s < 8000:
        t = tiny[(short(n), prev_name)]
        t[0] += 1; t[1] += e - s; t[2] += gap
    prev_end, prev_name = max(prev_end, e), short(n)
print("# idle gaps in the step: %.3f ms" % (gap_total / 1e6))
print("# launches under 8 us by (kernel <- previous kernel): count, summed us, summed gap in front")
for k, v in sorted(tiny.items(), key=lambda kv: -kv[1][1]):
    print("#  %3d  %7.1f us  gaps %6.1f us   %s  <-  %s" % (v[0], v[1] / 1e3, v[2] / 1e3, k[0], k[1]))
print("# total under 8 us: %d launches, %.3f ms" % (sum(v[0] for v in tiny.values()), sum(v[1] for v in tiny.values()) / 1e6))
