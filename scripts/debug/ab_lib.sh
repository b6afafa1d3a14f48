# A/B two builds of the library inside one gpurun call: bash scripts/debug/ab_lib.sh avex_amd/lib/libavexhip_base.so
run() { python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
AVEX_AMD_LIB=$PWD/$1 run "base "
run "new  "
done
