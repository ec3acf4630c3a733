#!/bin/bash
# One gpurun call of round 4 (same conventions as tools/r3_call.sh): bash tools/r4_call.sh <tag> [steps...]
# steps: ssltests sslbench sslbench16 sslops sslstats
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=${1:-r4x}; shift
STEPS=${*:-ssltests sslbench}
O=gpurun_out/$TAG
mkdir -p $O
guard() { local rc=$1; if [ $rc -ge 124 ]; then echo "step ended with $rc: stopping"; exit $rc; fi; }
for s in $STEPS; do
  echo "== $s $(date +%T)"
  case $s in
    ssltests)
      timeout -k 10 900 python -m pytest tests/test_ssl_step_gpu.py tests/test_ssl_blocks_gpu.py tests/test_ssl_dist_gpu.py tests/test_ssl_losses_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -s -p no:cacheprovider > $O/pytest_ssl.txt 2>&1; rc=$?; grep -E "^G12|passed|failed|Error|error" $O/pytest_ssl.txt | tail -12; guard $rc ;;
    sslbench)
      timeout -k 10 600 python3 tools/ssl_bench.py > $O/ssl_bench.json 2> $O/ssl_bench.err; rc=$?; cat $O/ssl_bench.json; tail -3 $O/ssl_bench.err; guard $rc ;;
    sslbench16)
      timeout -k 10 600 python3 tools/ssl_bench.py --dtype fp16 > $O/ssl_bench_fp16.json 2> $O/ssl_bench_fp16.err; rc=$?; cat $O/ssl_bench_fp16.json; tail -3 $O/ssl_bench_fp16.err; guard $rc ;;
    sslops)
      timeout -k 10 600 python3 tools/ssl_torch_ops.py > $O/ssl_torch_ops.md 2> $O/ssl_torch_ops.err; rc=$?; head -50 $O/ssl_torch_ops.md; tail -3 $O/ssl_torch_ops.err; guard $rc ;;
    sslstats)
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ssltrace -- python3 tools/ssl_bench.py --steps 4 --warmup 2 > $O/ssl_trace.log 2>&1; rc=$?; guard $rc
      python3 tools/summarize_prof.py $(find $O/ssltrace -name "*kernel_stats.csv" | head -1) 6 > $O/ssl_kernel_stats.md; rm -rf $O/ssltrace; head -45 $O/ssl_kernel_stats.md ;;
    tests)
      timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; rc=$?; tail -5 $O/pytest_gpu.txt; guard $rc ;;
    bench)
      timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 3500 $O/bench.json; tail -3 $O/bench.err; guard $rc ;;
    stats)
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg > $O/trace_bench.log 2>&1; rc=$?; guard $rc
      python3 tools/summarize_prof.py $(find $O/trace -name "*kernel_stats.csv" | head -1) 14 --cfg2 > $O/kernel_stats.md; cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/trace; head -40 $O/kernel_stats.md ;;
  esac
done
echo "== done $(date +%T)"
