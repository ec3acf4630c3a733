#!/bin/bash
# Where the column-masked dW kernel's time goes, in the step (cold operands, the batched launches): rocprofv3 kernel stats of a short
# bench run on the product library and on the three ablation libraries of tools/build_ablations.sh (DW_NOMMA / DW_NODMA / DW_NOEPI).
# bash tools/dw_ablate.sh [extra bench args, e.g. --backbone vit_large --batch 256 --patch 14 --partial-size 256]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/dw_ablate; mkdir -p $O
for v in product DW_NOMMA DW_NODMA DW_NOEPI; do
  if [ $v = product ]; then unset APLA_LIB; else export APLA_LIB=apla_amd/build/exp/libapla_$v.so; fi
  rm -rf $O/t_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg "$@" > $O/$v.log 2>&1 || exit 1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "proj_dw" in r["Name"]:
        print(f"{sys.argv[2]:10s} {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg us {float(r['AverageNs'])/1e3:8.1f}")
PY
  rm -rf $O/t_$v
done
