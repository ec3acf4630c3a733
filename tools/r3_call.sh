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
    gemm)
      timeout -k 10 600 python3 tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; rc=$?; cat $O/gemm_bench.txt; guard $rc ;;
  esac
done
echo "== done $(date +%T)"
