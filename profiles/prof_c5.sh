#!/bin/bash
# rocprofv3 kernel stats of the C5 (bf16, 4 modalities, 5 classes) step alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5 -o c5 -- python profiles/step_only.py 128 bf16 4 5 > gpurun_out/c5/c5.log 2>&1
rm -f gpurun_out/c5/*_kernel_trace.csv
tail -1 gpurun_out/c5/c5.log
