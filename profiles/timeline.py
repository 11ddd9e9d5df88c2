"""Timeline of the LAST training step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
every dispatch in start order with its duration and the idle gap before it, plus totals per kernel family.
Usage: python profiles/timeline.py <kernel_trace.csv> [steps_in_run] > timeline.txt"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


def main(path, steps):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size", "")) for r in csv.DictReader(open(path))]
    rows.sort()
    # one step = from one adam_kernel to the next
    adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    a, b = adam[-2] + 1, adam[-1] + 1
    step = rows[a:b]
    t0 = rows[a - 1][1]
    span = step[-1][1] - t0
    busy = sum(e - s for s, e, _, _ in step)
    gaps = 0
    prev = t0
    fam = defaultdict(lambda: [0, 0, 0])
    print("# one step: %d dispatches, span %.3f ms, sum of durations %.3f ms" % (len(step), span / 1e6, busy / 1e6))
    for s, e, n, g in step:
        gap = s - prev
        gaps += max(gap, 0)
        f = fam[n]
        f[0] += 1; f[1] += e - s; f[2] += max(gap, 0)
        print("%9.1f us  +%6.1f us gap  %8.1f us  %s grid=%s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, n, g))
        prev = max(prev, e)
    print("# idle gaps total %.3f ms" % (gaps / 1e6))
    print("# per kernel: calls, busy ms, gap-before ms")
    for n, (c, d, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print("# %4d %8.3f %8.3f  %s" % (c, d / 1e6, g / 1e6, n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
