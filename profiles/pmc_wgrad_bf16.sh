cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE"; do
  n=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_wg/$n -- python profiles/bench_one.py wgrad bf16 128 32 16 3 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_wg/*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:50]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in acc:
            if 'wgrad5_bf16' in k:
                print(k, {c: v / cnt[(k, c)] for c, v in acc[k].items()})
PY
