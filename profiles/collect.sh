#!/bin/bash
# Collects the evidence of one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <outdir under gpurun_out>
# 1. the bench line exactly as the driver runs it (incl. c3_f32x3, c5_bf16, c2_64cube_b2 and cpu_baseline),
# 2. rocprofv3 --kernel-trace --stats ONE LEG PER RUN (round 5, VERDICT r4 #5: the persistent-grid kernels have one grid whatever the
#    problem, so a run that mixes the 128^3 headline with the 64^3 B=2 leg averaged two shapes into one row): fp32 headline only,
#    fp32_split3 only, C5 (bf16) only, configs[1] (64^3, B=2) only -> <leg>_kernel_stats.csv,
# 3. separate --pmc FETCH_SIZE / WRITE_SIZE passes per leg (never combined with other trace domains),
# 4. the product training loop (image2label.train(), PCIe-inclusive),
# 5. the per-layer tables of the 5^3 / 2^3 launches (one per leg),
# 6. the streaming batch-norm passes one kernel at a time (profiles/bench_bn.py).
# Every step is bounded by `timeout`; python is the program right after `--`.
OUT=gpurun_out/${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
[ -n "$LEGS" ] || timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
COMMON="--no-cpu-baseline --no-sustained --no-c5 --no-c2 --no-x3"
leg_args() {
  case $1 in
    fp32) echo "--gpus 1 --compute fp32 $COMMON" ;;
    x3)   echo "--gpus 1 --compute fp32_split3 $COMMON" ;;
    c5)   echo "--gpus 1 --compute bf16 --channels 4 --classes 5 $COMMON" ;;
    c2)   echo "--gpus 1 --patch 64 --batch 2 $COMMON" ;;
  esac
}
for LEG in ${LEGS:-fp32 x3 c5 c2}; do
  A=$(leg_args $LEG)
  BENCH_NO_HBM_TABLE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${LEG} -- python bench.py $A --steps 20 --warmup 5 > $OUT/${LEG}_stats.log 2>&1
  if [ $LEG != c2 ]; then
    BENCH_NO_HBM_TABLE=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o ${LEG}_fetch -- python bench.py $A --steps 2 --warmup 1 > $OUT/${LEG}_fetch.log 2>&1
    BENCH_NO_HBM_TABLE=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o ${LEG}_write -- python bench.py $A --steps 2 --warmup 1 > $OUT/${LEG}_write.log 2>&1
  fi
  BENCH_NO_HBM_TABLE=1 BENCH_KERNEL_TABLE=1 timeout 600 python bench.py $A --steps 10 > /dev/null 2> $OUT/${LEG}_layer_table.err
  grep "^#" $OUT/${LEG}_layer_table.err > $OUT/${LEG}_layer_table.txt
done
[ -n "$LEGS" ] && exit 0      # (LEGS="fp32" bash profiles/collect.sh <dir>: re-run only those legs' rocprofv3 passes)
timeout 300 python profiles/train_loop_bench.py > $OUT/train_loop.json 2> $OUT/train_loop.err
timeout 300 python profiles/train_loop_bench.py 128 bf16 4 5 > $OUT/train_loop_c5.json 2> $OUT/train_loop_c5.err
timeout 300 python profiles/train_loop_bench.py 128 fp32_split3 1 2 > $OUT/train_loop_x3.json 2> $OUT/train_loop_x3.err
timeout 300 python profiles/bench_bn.py 200 > $OUT/bn_passes.txt 2> $OUT/bn_passes.err
rm -f $OUT/*_fetch_kernel_trace.csv $OUT/*_write_kernel_trace.csv      # (the counter-free traces stay: per-dispatch family rows, summarize.py)
ls -la $OUT | head -60
cut -c1-700 $OUT/bench_line.json
