cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in 0 1; do for z in "" 1; do
  printf "C16PP=%s ZERO=%s  " $v "$z"
  env VNET_BF16_C16PP=$v BENCH_ZERO=$z timeout 120 python profiles/bench_one.py conv bf16 128 16 16 50 2>&1 | tail -1
done; done; done
