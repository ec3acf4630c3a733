#!/bin/bash
# Round 6: one gpurun call made of named steps (tools/r6_call.sh step1 step2 ...); everything is written under gpurun_out/r6/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r6
mkdir -p $OUT
EXP=apla_amd/build/exp
for step in "$@"; do
  echo "=== $step $(date +%T)"
  case $step in
    lw_bench)   # loader-wave kernel (18; 1018 = two loader waves) against the product dispatch, images + rotating buffers as in the step
      GEMM_ONLY="fc1+gelu,dfc2*g,fc1+gelu_fwd" GEMM_IMAGES=1 GEMM_ROTATE=4 GEMM_VARIANTS=0,18,1018 timeout -k 10 300 python3 tools/gemm_bench.py > $OUT/lw_bench.txt 2>&1 || { tail -20 $OUT/lw_bench.txt; exit 1; }
      cat $OUT/lw_bench.txt ;;
    lw_bench_plain)
      GEMM_ONLY="qkv,proj,fc2,dfc1,dqkv" GEMM_ROTATE=4 GEMM_VARIANTS=0,18,1018 timeout -k 10 300 python3 tools/gemm_bench.py > $OUT/lw_bench_plain.txt 2>&1 || { tail -20 $OUT/lw_bench_plain.txt; exit 1; }
      cat $OUT/lw_bench_plain.txt ;;
    nopk_bench)  # the same sources compiled without packed fp32 vector operations
      APLA_LIB=$EXP/libapla_NOPK.so GEMM_ONLY="fc1+gelu,dfc2*g,fc1+gelu_fwd" GEMM_IMAGES=1 GEMM_ROTATE=4 GEMM_VARIANTS=0,18,1018 timeout -k 10 300 python3 tools/gemm_bench.py > $OUT/nopk_bench.txt 2>&1 || { tail -20 $OUT/nopk_bench.txt; exit 1; }
      cat $OUT/nopk_bench.txt ;;
    nt_ablations)  # clock-stamped ablation builds of the 4-wave persistent kernel: what the K loop costs without LDS-DMA issue / fragment reads / GELU
      for n in CLOCK CLOCK_NODMA CLOCK_NOREAD CLOCK_NODMA_NOREAD CLOCK_NOGELU CLOCK_NODMA_NOREAD_NOGELU; do
        echo "| **$n** | | | | | | | |" >> $OUT/nt_ablations.md
        APLA_LIB=$EXP/libapla_NT_$n.so CLOCK_ONLY="fc1,dfc2" CLOCK_KERNEL="" timeout -k 10 200 python3 tools/gemm_clock.py 0.5 2>&1 | grep "^|" | grep -v "launch |\|---" >> $OUT/nt_ablations.md || exit 1
      done
      cat $OUT/nt_ablations.md ;;
    ngrp_pp2)   # tile walk of the ping-pong kernel on the N = 768 launches (fc2, dfc1, dqkv): n-tiles per column group from APLA_NGRP (NGRP build)
      for G in -1 0 1 2 3 -1; do
        echo "--- APLA_NGRP=$G" >> $OUT/ngrp_pp2.txt
        APLA_NGRP=$G APLA_LIB=$EXP/libapla_NGRP.so GEMM_ONLY="fc2,dfc1,dqkv,qkv" GEMM_ROTATE=4 GEMM_VARIANTS=109 timeout -k 10 120 python3 tools/gemm_bench.py 2>&1 | grep "us" >> $OUT/ngrp_pp2.txt || exit 1
      done
      cat $OUT/ngrp_pp2.txt ;;
    ngrp_cfg3)   # the same sweep at config 3's shapes (ViT-L/14, bs 256: M = 65 792, D = 1024, F = 4096): four column tiles at N = 1024
      for G in -1 0 2 4 -1; do
        echo "--- APLA_NGRP=$G" >> $OUT/ngrp_cfg3.txt
        APLA_NGRP=$G APLA_LIB=$EXP/libapla_NGRP.so GEMM_M=65792 GEMM_SHAPES="fc2:1024:4096,dqkv:1024:3072,qkv:3072:1024,proj:1024:1024" GEMM_ROTATE=2 GEMM_VARIANTS=109 timeout -k 10 200 python3 tools/gemm_bench.py 2>&1 | grep "us" >> $OUT/ngrp_cfg3.txt || exit 1
      done
      cat $OUT/ngrp_cfg3.txt
      echo "--- two-output GELU at config 3: ping-pong (9) vs 4-wave persistent (15)" >> $OUT/ngrp_cfg3.txt
      GEMM_M=65792 GEMM_SHAPES="fc1:4096:1024:gelu,dfc2:4096:1024:mul" GEMM_IMAGES=1 GEMM_ROTATE=2 GEMM_VARIANTS=0,15 timeout -k 10 200 python3 tools/gemm_bench.py 2>&1 | grep "us" | tee -a $OUT/ngrp_cfg3.txt ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
