run() { python bench.py --steps 10 --warmup 3 $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run "chunk256 1 stream " ""
AVEX_AMD_STREAMS=2 run "chunk128 2 streams" "--chunk 128"
AVEX_AMD_STREAMS=2 run "chunk64  2 streams" "--chunk 64"
AVEX_AMD_STREAMS=4 run "chunk64  4 streams" "--chunk 64"
AVEX_AMD_STREAMS=3 run "chunk86  3 streams" "--chunk 86"
done
