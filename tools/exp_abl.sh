#!/bin/bash
# Experiment: A/B of the ping-pong GEMM against the previous build (apla_amd/build/exp/libapla_old.so)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/abl; mkdir -p $O
python3 -m pytest tests/test_kernels_gpu.py -q -x -k gemm -p no:cacheprovider 2>&1 | tail -3 | tee $O/log.txt
export GEMM_ONLY="qkv,fc1+gelu,dfc1,dproj" GEMM_VARIANTS=9
for v in old "" old ""; do
  echo "== build ${v:-product}" | tee -a $O/log.txt
  if [ -z "$v" ]; then python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt
  else APLA_LIB=$PWD/apla_amd/build/exp/libapla_$v.so python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/log.txt; fi
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee -a $O/log.txt
