#!/bin/bash
# Same-box A/B of the self-supervised iteration: .ab_prev/ holds `git archive <earlier commit>` built in place; tools/ssl_bench.py of both
# trees back to back, alternating, on the box of ONE gpurun call.  bash tools/ab_ssl.sh [steps]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/ab_ssl; mkdir -p $O
S=${1:-20}
for rep in 1 2 3; do
  (cd .ab_prev && timeout -k 10 300 python3 tools/ssl_bench.py --steps $S --warmup 4 > ../$O/prev_$rep.json 2> ../$O/prev_$rep.err) || exit 1
  timeout -k 10 300 python3 tools/ssl_bench.py --steps $S --warmup 4 > $O/new_$rep.json 2> $O/new_$rep.err || exit 1
done
python3 - <<PY
import json
for tag in ("prev", "new"):
    v = [json.loads(open(f"$O/{tag}_{i}.json").read().strip().splitlines()[-1]) for i in (1, 2, 3)]
    print(tag, "ms/iteration", [d["ms_per_step"] for d in v], "loss", [d["loss"] for d in v])
PY
