#!/bin/bash
# per-launch durations of the streaming batch-norm kernels in one 128^3 step (rocprofv3 kernel trace)
#   bash profiles/trace_bn.sh [fp32 1 2 | bf16 4 5]
MODE=${1:-fp32}; CIN=${2:-1}; K=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bntrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bntrace -o t -- python profiles/step_only.py 128 $MODE $CIN $K > gpurun_out/bntrace/log.txt 2>&1
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open('gpurun_out/bntrace/t_kernel_trace.csv')))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    m = re.search(r'(bn_act_\w+|bn_stats_\w+|bn_finalize_kernel|sum_finalize_kernel|splitk_reduce_b16_kernel|splitk_reduce_kernel|wgrad_reduce_batched_kernel|adam_kernel|pack_batched_kernel|head_\w+_kernel|conv_kernel<[12], [12][^>]*>|wgrad_kernel<2[^>]*>|conv2_\w+_kernel<[^>]*>)', n)
    if m:
        d[m.group(1)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
steps = 132
for k, v in sorted(d.items()):
    per = max(1, round(len(v) / steps))
    # launches come in the same order every step: average position by position over the steps
    avg = [sum(v[i::per][:steps]) / len(v[i::per][:steps]) for i in range(per)]
    print('%-60s %2d per step (us, launch order): %s' % (k, per, ' '.join('%.1f' % t for t in avg)))
PY
rm -f gpurun_out/bntrace/t_kernel_trace.csv
