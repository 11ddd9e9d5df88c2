#!/bin/bash
# Collects the evidence of one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <outdir under gpurun_out>
# 1. the bench line exactly as the driver runs it (incl. c5_bf16 and cpu_baseline),
# 2. rocprofv3 --kernel-trace --stats of the same command (without the CPU leg, which launches no kernels),
# 3./4. separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with other trace domains),
# 5. the product training loop (image2label.train(), PCIe-inclusive),
# 6. the per-layer table of the 5^3 / 2^3 launches (fp32 net, then the C5 bf16 net),
# 7. the streaming batch-norm passes one kernel at a time (profiles/bench_bn.py).
# Every step is bounded by `timeout`; python is the program right after `--`.
OUT=gpurun_out/${1:-prof}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sustained > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sustained > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sustained > $OUT/write.log 2>&1
timeout 300 python profiles/train_loop_bench.py > $OUT/train_loop.json 2> $OUT/train_loop.err
timeout 300 python profiles/train_loop_bench.py 128 bf16 4 5 > $OUT/train_loop_c5.json 2> $OUT/train_loop_c5.err
BENCH_KERNEL_TABLE=1 timeout 600 python bench.py --gpus 1 --steps 10 --no-cpu-baseline --no-sustained > /dev/null 2> $OUT/layer_table.err
grep "^#" $OUT/layer_table.err > $OUT/layer_table.txt
timeout 300 python profiles/bench_bn.py 200 > $OUT/bn_passes.txt 2> $OUT/bn_passes.err
rm -f $OUT/*_kernel_trace.csv $OUT/fetch_counter_collection.csv.bak
ls -la $OUT | head -30
cut -c1-700 $OUT/bench_line.json
