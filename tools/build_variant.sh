#!/bin/bash
# Build a diagnostic variant of the library: tools/build_variant.sh NAME SOURCE "-DDEFINES"  ->  apla_amd/build/exp/libapla_NAME.so
# (one source recompiled with the defines, linked with the product objects; run after `python -m apla_amd.build`)
set -e
cd "$(dirname "$0")/../apla_amd/build"
mkdir -p exp
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -w $3 -c ../csrc/$2.hip -o exp/$2_$1.o
objs=""
for o in errors gemm_nt gemm_pp2 gemm_w4 gemm_tp gemm_lw gemm_small layernorm attention apla_dw optim misc; do
  if [ $o = $2 ]; then objs="$objs exp/$2_$1.o"; else objs="$objs $o.o"; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o exp/libapla_$1.so $objs
echo "built apla_amd/build/exp/libapla_$1.so"
