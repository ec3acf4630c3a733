#!/bin/bash
# One gpurun call of round 3: GPU tests, bench line, one-step kernel trace, contention probe, kernel stats of the bench command.
# usage (from the repo root on the GPU box): bash tools/r3_call.sh <tag> [steps...]   steps: tests bench trace probe stats gemm
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=${1:-r3x}; shift
STEPS=${*:-tests bench trace probe stats}
O=gpurun_out/$TAG
mkdir -p $O
guard() {  # stop the whole call after a timeout / kill (a hung GPU step must not be followed by another one)
  local rc=$1
  if [ $rc -ge 124 ]; then echo "step ended with $rc: stopping"; exit $rc; fi
}
for s in $STEPS; do
  echo "== $s $(date +%T)"
  case $s in
    tests)
      timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; rc=$?; tail -5 $O/pytest_gpu.txt; guard $rc ;;
    testsall)
      timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; rc=$?; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.txt | tail -30; guard $rc ;;
    bench)
      timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 2500 $O/bench.json; tail -3 $O/bench.err; guard $rc ;;
    trace)
      timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/steptrace -- python3 tools/step_trace.py run > $O/steptrace.log 2>&1; rc=$?; guard $rc
      python3 tools/step_trace.py parse $(find $O/steptrace -name "*kernel_trace.csv" | head -1) > $O/step_launches.md 2>> $O/steptrace.log; head -5 $O/step_launches.md; rm -rf $O/steptrace ;;
    probe)
      timeout -k 10 900 python3 tools/contention_probe.py 150 > $O/contention.md 2> $O/contention.err; rc=$?; cat $O/contention.md; tail -3 $O/contention.err; guard $rc ;;
    stats)
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe > $O/trace_bench.log 2>&1; rc=$?; guard $rc
      python3 tools/summarize_prof.py $(find $O/trace -name "*kernel_stats.csv" | head -1) 14 --cfg2 > $O/kernel_stats.md; cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/trace; head -40 $O/kernel_stats.md ;;
    ssl)
      timeout -k 10 600 python3 tools/ssl_bench.py > $O/ssl_bench.json 2> $O/ssl_bench.err; rc=$?; cat $O/ssl_bench.json; tail -3 $O/ssl_bench.err; guard $rc
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ssltrace -- python3 tools/ssl_bench.py --steps 4 --warmup 2 > $O/ssl_trace.log 2>&1; rc=$?; guard $rc
      python3 tools/summarize_prof.py $(find $O/ssltrace -name "*kernel_stats.csv" | head -1) 6 > $O/ssl_kernel_stats.md; rm -rf $O/ssltrace; head -45 $O/ssl_kernel_stats.md ;;
    ssltests)
      timeout -k 10 900 python -m pytest tests/test_ssl_step_gpu.py tests/test_ssl_blocks_gpu.py tests/test_ssl_dist_gpu.py tests/test_ssl_losses_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_ssl.txt 2>&1; rc=$?; tail -5 $O/pytest_ssl.txt; guard $rc ;;
    cfg5)
      timeout -k 10 900 python bench.py --backbone vit_giant --img 518 --patch 14 --batch 32 --partial-size 512 --dtype fp16 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe > $O/bench_cfg5.json 2> $O/bench_cfg5.err; rc=$?; tail -c 1500 $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err; guard $rc ;;
    cfg3)
      timeout -k 10 900 python bench.py --backbone vit_large --patch 14 --batch 256 --partial-size 256 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-peak-probe > $O/bench_cfg3.json 2> $O/bench_cfg3.err; rc=$?; tail -c 1500 $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err; guard $rc ;;
    cfg5stats)
      timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -- python3 bench.py --backbone vit_giant --img 518 --patch 14 --batch 32 --partial-size 512 --dtype fp16 --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe > $O/trace_cfg5.log 2>&1; rc=$?; guard $rc
      python3 tools/summarize_prof.py $(find $O/trace5 -name "*kernel_stats.csv" | head -1) 9 > $O/kernel_stats_cfg5.md; rm -rf $O/trace5; head -30 $O/kernel_stats_cfg5.md ;;
    cfg3stats)
      timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -- python3 bench.py --backbone vit_large --patch 14 --batch 256 --partial-size 256 --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe > $O/trace_cfg3.log 2>&1; rc=$?; guard $rc
      python3 tools/summarize_prof.py $(find $O/trace3 -name "*kernel_stats.csv" | head -1) 10 > $O/kernel_stats_cfg3.md; rm -rf $O/trace3; head -30 $O/kernel_stats_cfg3.md ;;
    gemm)
      timeout -k 10 600 python3 tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; rc=$?; cat $O/gemm_bench.txt; guard $rc ;;
  esac
done
echo "== done $(date +%T)"
