#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python3 tools/dw_bench.py 2>&1 | grep -v amdgpu.ids
for s in 8 16 24 32 40 48; do APLA_DW_SLABS=$s APLA_LIB=$PWD/apla_amd/build/exp/libapla_DWSLABS.so python3 tools/dw_bench.py 2>&1 | grep proj_dw; done
