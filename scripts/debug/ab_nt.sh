run() { python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
AVEX_AMD_GEMM_NT=0 run "no-nt            "
AVEX_AMD_GEMM_NT=1 run "nt-stores        "
AVEX_AMD_GEMM_NT=5 run "nt-only-N>768    "
AVEX_AMD_GEMM_NT=1 AVEX_AMD_ATT_DEBUG=8 run "nt+att-nt        "
done
