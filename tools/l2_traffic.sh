#!/bin/bash
# Bytes fetched from beyond L2 (FETCH_SIZE) and launch time of one GEMM shape as a function of the tile walk's column-group width
# (APLA_NGRP, NGRPALL build of tools/build_ablations.sh).  usage on the GPU box: bash tools/l2_traffic.sh <out-dir>
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=${1:-gpurun_out/l2}
mkdir -p $O
export APLA_LIB=$(pwd)/apla_amd/build/exp/libapla_NGRPALL.so
run() {  # label N K epi variant ngrp kernel-substring
  local tag=run_$2_$3_$4_$5_ngrp$6
  APLA_NGRP=$6 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$tag -- python3 tools/gemm_one.py $2 $3 $4 $5 8 > $O/last.log 2>&1
  rc=$?; if [ $rc -ge 124 ]; then echo "timeout in $tag"; exit $rc; fi
  local f=$(python3 tools/pmc_summary.py $O/$tag "$7" | grep FETCH_SIZE | sed 's/.*avg=//')
  local t=$(python3 - <<PY
import csv, glob
rows = [r for f in glob.glob("$O/$tag/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "$7" in r["Kernel_Name"]]
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"{d[len(d) // 2] / 1e3:.1f}" if d else "nan")
PY
)
  echo "| $1 N=$2 K=$3 | $6 | $f | $t |"
  rm -rf $O/$tag
}
echo "| launch | APLA_NGRP (n-tiles per column group; -1 = the product's choice, 0 = none) | FETCH_SIZE KiB as reported (x2 on gfx950) | median launch us (under the profiler) |"
echo "|---|---:|---:|---:|"
for g in -1 4 6 8 12 0; do run "fc1+GELU (4-wave 128-wide)" 3072 768 1 4 $g gemm_persist_kernel; done
for g in -1 2 3 5 0; do run "qkv (wide 4-wave, W image)" 2304 768 0 116 $g gemm_w4_kernel; done
for g in -1 2 3 5 0; do run "qkv (ping-pong, W image)" 2304 768 0 109 $g gemm_pp2_kernel; done
for g in -1 3 6 0; do run "fc1 plain (wide 4-wave, W image)" 3072 768 0 116 $g gemm_w4_kernel; done
