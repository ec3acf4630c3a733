"""bench.py's output contract (one JSON line on stdout with the fields the round driver and the judge read), run as the
driver runs it: a child process, default workload (BASELINE config 2), few steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--cpu-sample-bs", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "ViT-B/16" in d["metric"] and "bs=128" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 1000 and abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - 2 * 25216 * 768 * 3072 / (rf["kernel_ms"] * 1e-3) / 1e12) < 0.01 * rf["achieved"]   # algorithmic FLOP / in-place duration
    assert rf["traffic"] is None or rf["traffic"] > rf["algorithmic_bytes"] * 0.9
    assert rf["traffic"] is None or (rf["traffic_source"]["stored"] is True and rf["traffic_source"]["file"].startswith("profiles/"))
    assert rf["peak_measured"] is None or 1000.0 < rf["peak_measured"] < 2600.0     # tools/mfma_peak on this GPU
    assert rf["step_frac_executed"] <= rf["step_frac"] and rf["step_gflop_per_image_executed"] < rf["step_gflop_per_image"]
    assert d["ms_per_step_median"] > 0 and d["ranks_seen"] == 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "images/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["steps"] >= 5 and cb["statistic"] == "median" and cb["cpus_usable"] >= cb["cores"] and str(cb["cores"]) in cb["by_threads"]
    par = d["parity"]      # config 1 through the same build and dtype, against the reference's CPU logits
    assert par["dtype"] == "bf16" and par["tol_asserted"] == 8e-3 and par["logits_rel_vs_reference_cfg1"] < par["tol_asserted"]
    assert par["n_batches"] == 8 and par["logits_rel_mean_over_batches"] <= par["logits_rel_max_over_batches"] < par["tol_asserted"]
    assert abs(par["loss"] - par["loss_reference"]) < 5e-3      # the seed-0 batch's loss against the reference's loss on that batch
    p16, f16 = d["parity_fp16"], d["fp16"]      # the configuration that meets the north-star's 1e-3: parity AND throughput in the line
    # one record, one bound per statistic: the maximum over the batches is asserted at 1.2e-3 (the fp16 operand roundings alone give
    # 1.04e-3 on these batches, tests/test_fp16_gpu.py), the mean at the north-star's 1e-3
    assert p16["dtype"] == "fp16" and p16["tol_asserted"] == 1.2e-3 and p16["tol_asserted_on_the_mean"] == 1e-3 and p16["north_star_tol"] == 1e-3
    assert p16["logits_rel_mean_over_batches"] < p16["tol_asserted_on_the_mean"] and p16["logits_rel_max_over_batches"] < p16["tol_asserted"]
    assert abs(p16["loss"] - p16["loss_reference"]) < 1e-3
    pm = d["parity_benchmarked_model"]   # the benchmarked model shape itself (ViT-B/16, r = 192, C = 1000) on the CPU-baseline sample, both dtypes
    assert "vit_base/16" in pm["reference"] and "bs=2" in pm["reference"]
    assert pm["bf16"]["logits_rel_max"] < 1.2e-2 and pm["bf16"]["logits_rel_l2"] < 1e-2 and abs(pm["bf16"]["loss"] - pm["loss_reference"]) < 5e-3
    assert pm["fp16"]["logits_rel_max"] < 2.0e-3 and pm["fp16"]["logits_rel_l2"] < 1.5e-3 and abs(pm["fp16"]["loss"] - pm["loss_reference"]) < 1e-3
    assert pm["fp16"]["logits_rel_max"] < pm["bf16"]["logits_rel_max"]
    assert f16["steps"] == 3 and f16["ms_per_step"] > 0 and abs(f16["images_per_sec"] - 128 / (f16["ms_per_step"] * 1e-3)) < 0.01 * f16["images_per_sec"]
    su = d["sustained"]                          # the steady-state figure of the same command: a stored record with its provenance
    assert su is None or (su["stored"] is True and su["file"].startswith("profiles/") and su["steps"] >= 1000 and su["ms_per_step"] > 0.9 * d["ms_per_step"] * 0.8)
    rk = d["ranks"]                              # a one-rank run has no exchange: nothing waited for
    assert rk["ms_per_step"] == [rk["ms_per_step_min"]] == [rk["ms_per_step_max"]] and rk["exchange_wait_ms"] == [None] and rk["exchange_chunks"] == 0
    pw = d["power"]                              # socket telemetry of the timed steps (null only where the driver's hwmon files are missing)
    assert len(rk["socket_w"]) == 1
    if pw is not None:
        assert pw["samples"] >= 1 and 100.0 < pw["mean_w"] <= pw["max_w"] <= 1.1 * pw["cap_w"] and 500 <= pw["sclk_mhz"] <= 2600
        assert rk["socket_w"] == [pw["mean_w"]]


def test_bench_reports_the_exchange_wait_on_the_forced_exchange_path():
    """APLA_FORCE_EXCHANGE=1: the world > 1 code path (four graph segments, an RCCL all-reduce per gradient chunk on the side stream,
    reserved CUs) on ONE rank — the bench line must then carry what a scaling run needs to explain itself: the time the compute stream
    stood at GradExchanger.wait(), every rank's own step time, and the RCCL / CU budget in force."""
    env = dict(os.environ, APLA_FORCE_EXCHANGE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                        "--no-parity", "--no-peak-probe", "--no-fp16-leg"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    rk = d["ranks"]
    assert rk["exchange_chunks"] == 4 and rk["reserved_cus"] == 8 and len(rk["exchange_wait_ms"]) == 1
    assert rk["exchange_wait_ms"][0] is not None and 0.0 <= rk["exchange_wait_ms"][0] < d["ms_per_step"]
    assert rk["env"]["NCCL_MAX_NCHANNELS"] == "8" and rk["env"]["APLA_RESERVE_CUS"] == "8" and rk["env"]["APLA_FORCE_EXCHANGE"] == "1"
    assert abs(rk["ms_per_step"][0] - d["ms_per_step"]) < 0.2 * d["ms_per_step"]


@pytest.mark.parametrize("launcher", ["self", "torch.distributed.run"])
def test_bench_default_line_through_the_multi_rank_path_on_one_gpu(launcher):
    """The driver's scaling run is `bench.py --gpus N` with DEFAULT flags — parity checks on rank 0 included — and no run of it had ever
    executed with more than one rank (the other multi-rank tests pass --no-parity).  `--rehearse-on-one-gpu` puts every rank on cuda:0
    over gloo (RCCL refuses two ranks on one device) and leaves everything else as the real N > 1 path: rendezvous, per-rank batches,
    four graph segments with the gradient exchange between them, MAX over ranks, the per-rank arrays — and rank 0's parity engines, which
    must be rank-LOCAL (round 6: they were built with process_group=None = the default group, i.e. rank 0 alone would have issued
    collectives and the job would have hung).  Three ranks: a world size that is neither 1 nor 2."""
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rehearse-on-one-gpu", "--steps", "3", "--warmup", "2", "--no-peak-probe"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:       # the form the driver uses for N > 1: one rank per process under torch.distributed.run
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3 and d["config"]["global_batch"] == 3 * 128 and d["config"]["parallelism"] == "dp3"
    rk = d["ranks"]
    assert len(rk["ms_per_step"]) == 3 and len(rk["exchange_wait_ms"]) == 3 and rk["exchange_chunks"] == 4 and rk["reserved_cus"] == 8
    assert all(w is not None and w >= 0.0 for w in rk["exchange_wait_ms"])
    assert d["parity"]["logits_rel_max_over_batches"] < d["parity"]["tol_asserted"]            # rank 0's check ran, alone, and is right
    assert d["parity_fp16"]["logits_rel_max_over_batches"] < d["parity_fp16"]["tol_asserted"]
    assert "cpu_baseline" not in d and "fp16" not in d                                          # single-GPU legs only
    assert d["value"] > 0 and abs(d["value"] - 3 * 128 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]


def test_bench_selflaunch_two_ranks():
    """`python bench.py --gpus 2` starts its two ranks itself (reference: src/utils/launch.py:49-58).  With two
    GPUs it must print one line with n_gpus = 2 and ranks_seen = 2 (counted by an RCCL all-reduce).  On a one-GPU box rank 1
    has no device: the parent must then report the failure — stop rank 0, exit non-zero, print no JSON — instead of hanging."""
    import torch
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-parity", "--no-peak-probe"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(lines[0])
        assert len(lines) == 1 and d["n_gpus"] == 2 and d["ranks_seen"] == 2
    else:
        assert r.returncode != 0 and not lines and "rank 1 exited" in r.stderr
