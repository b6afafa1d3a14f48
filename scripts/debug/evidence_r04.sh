# round-4 evidence set (GPU box): bench + rocprof stats + PMC traffic, SQ counters per GEMM shape, tile-walk traffic, variable-length allocation trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout 300 python -m pytest $R/tests/test_gpu_overflow.py -x -q 2>&1 | tail -3 > $R/gpurun_out/r04m_overflow_tests.txt
bash $R/scripts/collect_profiles.sh r04m > $R/gpurun_out/r04m_collect.log 2>&1
bash $R/scripts/pmc_gemm_sq.sh r04m > $R/gpurun_out/r04m_gemm_sq.txt 2>&1
bash $R/scripts/pmc_walks.sh > $R/gpurun_out/r04m_walks.log 2>&1
cd /tmp && export TMPDIR=/tmp
for n in 1 40; do
  rm -rf $R/gpurun_out/varlen_$n
  rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/varlen_$n -- python3 $R/scripts/varlen_alloc.py $n > $R/gpurun_out/varlen_$n.log 2>&1
done
python3 - $R/gpurun_out <<'PY' > $R/gpurun_out/r04m_varlen_hiptrace.txt
import csv, glob, os, sys
root = sys.argv[1]
print("rocprofv3 --hip-trace --stats -- python3 scripts/varlen_alloc.py N   (40 forwards of 2 clips after a warm-up with the longest clip)")
for n in (1, 40):
    calls = {}
    for f in glob.glob(os.path.join(root, f"varlen_{n}", "**", "*hip_api_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            calls[r["Name"]] = int(r["Calls"])
    keys = ["hipMalloc", "hipFree", "hipMemcpy", "hipMemcpyAsync", "hipMallocAsync", "hipFreeAsync", "hipHostMalloc", "hipLaunchKernel", "hipExtModuleLaunchKernel", "hipStreamSynchronize", "hipDeviceSynchronize", "hipEventRecord", "hipStreamWaitEvent"]
    print(f"N = {n:2d} distinct lengths: " + "  ".join(f"{k} {calls.get(k, 0)}" for k in keys))
    print("   " + open(os.path.join(root, f"varlen_{n}.log")).read().strip().splitlines()[-1])
PY
python3 $R/scripts/parity_report.py 2>&1 | grep -v amdgpu > $R/gpurun_out/r04m_parity_report.json
cat $R/gpurun_out/r04m_overflow_tests.txt $R/gpurun_out/r04m_gemm_sq.txt $R/gpurun_out/r04m_walks.log $R/gpurun_out/r04m_varlen_hiptrace.txt; tail -5 $R/gpurun_out/r04m_collect.log
