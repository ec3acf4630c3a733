#!/bin/bash
# Same-box A/B of the config-2 step between the product library and another build of it (APLA_LIB=<path>, e.g. an ablation library of
# tools/build_ablations.sh): bench.py back to back, alternating, three times.  bash tools/ab_lib.sh <lib.so> [steps] [extra bench args]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
LIB=$1; S=${2:-100}; shift; shift
O=gpurun_out/ab_lib; mkdir -p $O
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps $S --warmup 10 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg "$@" > $O/base_$rep.json 2> $O/base_$rep.err || exit 1
  APLA_LIB=$LIB timeout -k 10 300 python3 bench.py --steps $S --warmup 10 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg "$@" > $O/alt_$rep.json 2> $O/alt_$rep.err || exit 1
done
python3 - <<PY
import json
for tag in ("base", "alt"):
    v = [json.loads(open(f"$O/{tag}_{i}.json").read().strip().splitlines()[-1]) for i in (1, 2, 3)]
    print(tag, "ms/step", [d["ms_per_step"] for d in v], "median-of-steps", [d.get("ms_per_step_median") for d in v])
PY
