run() { python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run "fold-on  v2      "
AVEX_AMD_LN_FOLD=0 run "fold-off v2      "
AVEX_AMD_LN_FOLD=0 AVEX_AMD_GEMM_VARIANT=5 run "fold-off v5      "
done
