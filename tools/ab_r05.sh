#!/bin/bash
# Same-box A/B of the previous round's code against this round's (boxes of the pool differ by +-3 %, more than a round's gain):
# .ab_r05/ holds `git archive 934c140` (round 5's last commit) built in place; both bench.py run back to back, alternating, on the
# box of ONE gpurun call.  bash tools/ab_r05.sh [steps]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/ab_r05; mkdir -p $O
S=${1:-100}
for rep in 1 2 3; do
  (cd .ab_r05 && python3 bench.py --steps $S --warmup 10 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg > ../$O/r05_$rep.json 2> ../$O/r05_$rep.err)
  python3 bench.py --steps $S --warmup 10 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg > $O/r06_$rep.json 2> $O/r06_$rep.err
done
python3 - <<PY
import json
for tag in ("r05", "r06"):
    v = [json.loads(open(f"$O/{tag}_{i}.json").read().strip().splitlines()[-1]) for i in (1, 2, 3)]
    print(tag, "ms/step", [d["ms_per_step"] for d in v], "median-of-steps", [d["ms_per_step_median"] for d in v], "W", [d["power"]["mean_w"] if d.get("power") else None for d in v])
PY
