#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM traffic counters for the bench workload.
# Usage: bash scripts/collect_profiles.sh <tag>     (outputs under gpurun_out/<tag>_*)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
# counters in their own passes (kernel-trace only), as the microarch guide prescribes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python3 $R/scripts/parse_traffic.py $R/gpurun_out ${TAG}
