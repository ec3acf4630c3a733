#!/bin/bash
# Diagnostic builds of the library for ablation timing (see DESIGN.md appendix C.1 "Where the GEMM time goes"): each variant compiles ONE
# source with a -DAPLA_ABL_* switch and links it with the product objects into apla_amd/build/exp/libapla_<variant>.so.
# Results of these builds are WRONG by construction; they exist to be timed (tools/gemm_bench.py / tools/dw_bench.py with
# APLA_LIB=<path>).  Run after `python -m apla_amd.build`.
set -e
cd "$(dirname "$0")/../apla_amd/build"
mkdir -p exp
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result"
OBJS="errors gemm_nt gemm_pp2 gemm_w4 gemm_tp gemm_lw gemm_small layernorm attention apla_dw optim misc"
build() {  # name source "defines"
  $HIPCC $FLAGS $3 -c ../csrc/$2.hip -o exp/$2_$1.o
  local objs=""
  for o in $OBJS; do if [ $o = $2 ]; then objs="$objs exp/$2_$1.o"; else objs="$objs $o.o"; fi; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o exp/libapla_$1.so $objs
  echo "built exp/libapla_$1.so"
}
if [ "$1" = "tp" ]; then build TPSTAMPS gemm_tp "-DAPLA_ABL_TPSTAMPS"; exit 0; fi   # tile-alternating kernel with role stamps only (tools/tp_stamps.py)
build TPSTAMPS gemm_tp "-DAPLA_ABL_TPSTAMPS"         # tile-alternating kernel: cycles per role and activity (tools/tp_stamps.py)
build NOSTORE gemm_pp2 "-DAPLA_ABL_NOSTORE"          # ping-pong GEMM epilogue computes but does not store
build NOGELU gemm_pp2 "-DAPLA_ABL_NOGELU"            # GELU epilogue without the GELU arithmetic (two stores of the accumulators)
build NOREAD gemm_pp2 "-DAPLA_ABL_NOREAD"            # K loop without LDS fragment reads
build NODMA gemm_pp2 "-DAPLA_ABL_NODMA"              # K loop without LDS-DMA
build NOREADNODMA gemm_pp2 "-DAPLA_ABL_NOREAD -DAPLA_ABL_NODMA"   # MFMA + barriers only: the structure's floor
build SAMEK gemm_pp2 "-DAPLA_ABL_SAMEK"              # LDS-DMA always from the k = 0 slice (cache-resident source)
build DWRING3 apla_dw "-DAPLA_ABL_DWRING3"           # dW kernel with three stages and one workgroup per CU at every width (round 5's form; results stay right)
build DW_NOMMA apla_dw "-DAPLA_ABL_DW_NOMMA"         # dW kernel: LDS-DMA, waits and barriers only (no fragment reads, no products)
build DW_NODMA apla_dw "-DAPLA_ABL_DW_NODMA"         # dW kernel without its LDS-DMA (products on whatever the LDS holds)
build DW_NOEPI apla_dw "-DAPLA_ABL_DW_NOEPI"         # dW kernel without the partial-tile epilogue
build DWSLABS apla_dw "-DAPLA_ABL_DWSLABS"           # dW slab count from APLA_DW_SLABS
# attention backward (tools/attn_split.py under rocprofv3 --kernel-trace --stats, APLA_ATTN variant 1 = split kernels)
build ATT_NOEXP attention "-DAPLA_ABL_ATT_NOEXP"     # no transcendental in the softmax recompute
build ATT_NOS attention "-DAPLA_ABL_ATT_NOS"         # split kernels without the S / dP products
build ATT_NOTR attention "-DAPLA_ABL_ATT_NOTR"       # split kernels without the transposed reads and second-stage products
build LNGRID layernorm "-DAPLA_ABL_LNGRID"           # LayerNorm grid cap from APLA_LN_GRID
build NOW4STORE gemm_nt "-DAPLA_ABL_NOW4STORE"       # dispatch only (results stay right): short-K plain stores on the ping-pong kernel instead of the wide 4-wave kernel (tools/ab_lib.sh)
build NGRP gemm_pp2 "-DAPLA_ABL_NGRP"                # n-tiles per column group of the tile walk from APLA_NGRP (ping-pong kernel)
# tile-walk sweep over all three tiled GEMM kernels (tools/l2_traffic.sh): column-group width from APLA_NGRP
$HIPCC $FLAGS -DAPLA_ABL_NGRP -c ../csrc/gemm_nt.hip -o exp/gemm_nt_NGRPALL.o
$HIPCC $FLAGS -DAPLA_ABL_NGRP -c ../csrc/gemm_w4.hip -o exp/gemm_w4_NGRPALL.o
objs=""; for o in $OBJS; do case $o in gemm_nt) objs="$objs exp/gemm_nt_NGRPALL.o";; gemm_w4) objs="$objs exp/gemm_w4_NGRPALL.o";; gemm_pp2) objs="$objs exp/gemm_pp2_NGRP.o";; *) objs="$objs $o.o";; esac; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o exp/libapla_NGRPALL.so $objs
echo "built exp/libapla_NGRPALL.so"
# all three tiled GEMM kernels stamp their core clock per workgroup (tools/gemm_clock.py)
for f in gemm_nt gemm_w4 gemm_pp2; do $HIPCC $FLAGS -DAPLA_ABL_CLOCK -DAPLA_ABL_NGRP -c ../csrc/$f.hip -o exp/${f}_CLOCK.o; done
objs=""; for o in $OBJS; do case $o in gemm_nt|gemm_w4|gemm_pp2) objs="$objs exp/${o}_CLOCK.o";; *) objs="$objs $o.o";; esac; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o exp/libapla_CLOCK.so $objs
echo "built exp/libapla_CLOCK.so"
# the wide 4-wave kernel without LDS-DMA / without fragment reads / without both, with clock stamps (APLA_LIB=... python3 tools/gemm_clock.py)
for v in NODMA NOREAD "NODMA -DAPLA_ABL_NOREAD" SAMEK; do
  n=W4_$(echo $v | sed 's/ -DAPLA_ABL_//')
  $HIPCC $FLAGS -DAPLA_ABL_CLOCK -DAPLA_ABL_$v -c ../csrc/gemm_w4.hip -o exp/gemm_w4_$n.o
  objs=""; for o in $OBJS; do case $o in gemm_w4) objs="$objs exp/gemm_w4_$n.o";; gemm_nt|gemm_pp2) objs="$objs exp/${o}_CLOCK.o";; *) objs="$objs $o.o";; esac; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o exp/libapla_$n.so $objs
  echo "built exp/libapla_$n.so"
done
