#!/bin/bash
# Round measurement on the GPU box (run through gpurun from the repo root): GPU tests, bench line, rocprofv3 kernel stats
# of the same bench command, and the PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) for the dominant kernel.
# Everything lands in gpurun_out/round/; copy what should be judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/round
rm -rf $O && mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
tools/mfma_peak 1.0 > $O/mfma_peak.json 2>&1
[ -x tools/dma_probe ] && tools/dma_probe > $O/dma_probe.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg > $O/trace_bench.log 2>&1
python3 tools/summarize_prof.py $(find $O/trace -name "*kernel_stats.csv") 14 --cfg2 > $O/kernel_stats.md
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
# HBM counters of EVERY kernel of the step (separate passes, eager launches: a counter pass serialises the kernels anyway), then the
# machine-readable roofline table of the top kernels (tools/roofline_table.py; bench.py embeds profiles/roofline_kernels.json)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcb_$c -- python3 bench.py --steps 2 --warmup 1 --no-graphs --no-cpu-baseline --no-parity --no-peak-probe --no-fp16-leg > $O/pmcb_$c.log 2>&1
done
python3 tools/roofline_table.py $O/kernel_stats.csv 14 --pmc-fetch $O/pmcb_FETCH_SIZE --pmc-write $O/pmcb_WRITE_SIZE --top 14 --tag "${APLA_ROUND_TAG:-round 6}" > $O/roofline_kernels.json
rm -rf $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 tools/gemm_one.py 3072 768 1 4 6 > /dev/null 2>&1
  python3 tools/pmc_summary.py $O/pmc_$c "gemm_persist_kernel" > $O/pmc_$c.txt
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 tools/gemm_one.py 3072 768 1 4 6 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc_sq "gemm_persist_kernel" > $O/pmc_sq.txt
# the two persistent attention kernels of the step: traffic and issue counters (same conventions)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_attn_$c -- python3 tools/attn_one.py > /dev/null 2>&1
  python3 tools/pmc_summary.py $O/pmc_attn_$c "attn_" > $O/pmc_attn_$c.txt
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_attn_sq -- python3 tools/attn_one.py > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc_attn_sq "attn_" > $O/pmc_attn_sq.txt
rm -rf $O/pmc_attn_FETCH_SIZE $O/pmc_attn_WRITE_SIZE $O/pmc_attn_sq
python3 - <<PY
import re, json
def val(path, name):
    for l in open(path):
        if name in l:
            return float(l.split("avg=")[1])
d = {"kernel": "gemm_persist_kernel<GELU,bf16,5> M=25216 N=3072 K=768 (fc1+GELU launch)", "taken_at": "${APLA_ROUND_TAG:-round 6}",
     "FETCH_SIZE_KiB": val("$O/pmc_FETCH_SIZE.txt", "FETCH_SIZE"), "WRITE_SIZE_KiB": val("$O/pmc_WRITE_SIZE.txt", "WRITE_SIZE"),
     "note": "rocprofv3 --pmc, one counter per pass, averaged over 6 launches; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md HBM section)"}
try:   # the in-kernel clock record comes from its own tool (tools/gemm_clock.py, CLOCK build): carried over
    d["held_clock"] = json.load(open("profiles/pmc_dominant_kernel.json"))["held_clock"]
except (OSError, KeyError, ValueError):
    pass
json.dump(d, open("$O/pmc_dominant_kernel.json", "w"), indent=1)
print(d)
PY
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq
ls -la $O
