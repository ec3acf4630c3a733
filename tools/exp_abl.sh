#!/bin/bash
# Experiment: K-loop ablations of the ping-pong GEMM (diagnostic builds in apla_amd/build/exp)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/abl; mkdir -p $O
: > $O/log.txt
export GEMM_ONLY="qkv,dfc1" GEMM_VARIANTS=9
for v in "" SAMEK NODMA "" SAMEK NODMA; do
  echo "== build ${v:-product}" | tee -a $O/log.txt
  if [ -z "$v" ]; then python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt
  else APLA_LIB=$PWD/apla_amd/build/exp/libapla_$v.so python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt; fi
done
