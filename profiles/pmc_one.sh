#!/bin/bash
# PMC passes for ONE micro-benchmark problem: bash profiles/pmc_one.sh <tag> <bench_one args...>; each pass bounded by timeout.
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  n=$(echo $C | tr ' ' '_')
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$n -- python profiles/bench_one.py "$@" 3 > /dev/null 2>&1 || echo "pass $n failed/timeout"
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob(sys.argv[1] + '/*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
        for k in acc:
            if 'conv' in k or 'wgrad' in k:
                print(k, {c: round(v / len(disp[k])) for c, v in acc[k].items()})
PY
