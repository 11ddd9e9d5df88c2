#!/bin/bash
# Collects the evidence of one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <outdir under gpurun_out> [extra bench args]
# 1. the bench line (full default run incl. cpu_baseline), 2. rocprofv3 --kernel-trace --stats of the same command
#    (+ the same with VNET_PARAM_GRAD_STREAM=0: every kernel alone),
# 3./4. separate --pmc FETCH_SIZE / WRITE_SIZE passes (never combined with other trace domains).
# Every step is bounded by `timeout`; python is the program right after `--`.
OUT=gpurun_out/${1:-prof}; shift
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
timeout 900 python bench.py $EXTRA > $OUT/bench_line.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/stats.log 2>&1
# per-kernel durations without the filter-gradient stream overlapping the backward-data convs (every kernel runs alone)
VNET_PARAM_GRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o serial -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/serial.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/write.log 2>&1
ls -la $OUT | head -20
cat $OUT/bench_line.json | cut -c1-600
