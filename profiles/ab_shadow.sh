cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sh
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sh -o on -- python profiles/step_only.py 128 bf16 4 5 > gpurun_out/sh/on.log 2>&1
export VNET_BF16_SHADOW=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sh -o off -- python profiles/step_only.py 128 bf16 4 5 > gpurun_out/sh/off.log 2>&1
rm -f gpurun_out/sh/*_kernel_trace.csv
tail -2 gpurun_out/sh/on.log gpurun_out/sh/off.log
