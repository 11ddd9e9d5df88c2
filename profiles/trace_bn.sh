#!/bin/bash
# per-launch durations of the streaming batch-norm kernels in one fp32 128^3 step (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bntrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bntrace -o t -- python profiles/step_only.py 128 fp32 1 2 > gpurun_out/bntrace/log.txt 2>&1
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open('gpurun_out/bntrace/t_kernel_trace.csv')))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'bn_act' in n or 'adam' in n or 'pack_batched' in n:
        d[re.search(r'(bn_act_\w+|adam_kernel|pack_batched_kernel)', n).group(1)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v.sort(reverse=True)
    steps = 133
    print(k, len(v) // steps, 'per step; sorted per-step profile (us):', ' '.join('%.1f' % t for t in v[::steps][:34]))
PY
rm -f gpurun_out/bntrace/t_kernel_trace.csv
