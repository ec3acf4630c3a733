export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python3 -m pytest tests/test_kernels_gpu.py -q -x -k gemm -p no:cacheprovider 2>&1 | tail -2
export GEMM_ONLY="qkv,fc1+gelu,dfc1,dproj" GEMM_VARIANTS=9
for v in old "" old ""; do
  echo "== build ${v:-product}"
  if [ -z "$v" ]; then python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  else APLA_LIB=$PWD/apla_amd/build/exp/libapla_$v.so python3 tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids; fi
done
