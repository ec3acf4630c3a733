#!/usr/bin/env python3
"""What do resident foreign workgroups cost the fused step?  (a handle on RCCL / compute contention with one GPU.)

The data-parallel step launches an RCCL all-reduce per gradient chunk on a side stream while the backward continues
(apla_amd/dist.py; reference: DDP, defaults/wrappers.py:182-183).  RCCL's kernels occupy one CU per channel for as long as the
transfer lasts; the step's persistent GEMM / attention kernels want every CU at once.  With ONE GPU there is no peer to exchange
with, so this probe keeps k workgroups (256 threads, 16 KB of LDS: an RCCL-channel-sized footprint) resident for `usec`
microseconds on the side stream at each of the four points where the N > 1 step launches a collective (process group of one rank on
backend nccl, APLA_FORCE_EXCHANGE=1: the real four-segment path incl. the RCCL calls), and times the step:

    python3 tools/contention_probe.py [usec=150] > profiles/r03_contention.md

for k in {0, 4, 8, 16, 32, 64} and for the GEMM launches leaving 0 or k CUs free (ops.reserved_cus).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(k, reserve, usec, steps=30, warmup=8):
    import torch
    import torch.distributed as dist
    from bench import build_model
    from apla_amd import ops
    from apla_amd._lib import check, lib
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    os.environ["APLA_RESERVE_CUS"] = str(reserve)
    eng = AplaTrainEngine(build_model("vit_base", 192, 1000, 224, 16), 128, 224, optim=OptimConfig(lr=1e-4, weight_decay=1e-5, grad_clipping=1.0),
                          process_group=dist.group.WORLD)
    assert eng.exchanger.active and len(eng.seg_cuts) == 4 and eng.reserve_cus == reserve
    real = eng.exchanger.launch_chunk

    def launch_chunk(j):
        real(j)                                   # the one-rank RCCL all-reduce, as APLA_FORCE_EXCHANGE runs it
        if k > 0:
            with torch.cuda.stream(eng.exchanger._comm):
                check(lib().apla_probe_occupy(k, 256, 16384, usec, torch.cuda.current_stream().cuda_stream), "apla_probe_occupy")
    eng.exchanger.launch_chunk = launch_chunk
    g = torch.Generator(device="cuda").manual_seed(0)
    eng.set_batch(torch.randn(128, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (128,), device="cuda", generator=g))
    for _ in range(warmup):
        eng.train_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.train_step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    del eng
    torch.cuda.empty_cache()
    return ms


def main():
    import torch
    import torch.distributed as dist
    usec = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", APLA_FORCE_EXCHANGE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    print(f"# Step time with k foreign workgroups resident for {usec} us at each of the four exchange points (one MI355X, config 2)\n")
    print("| k resident workgroups | GEMMs on all 256 CUs: ms/step | GEMMs leave k CUs free: ms/step |\n|---:|---:|---:|")
    rec = []
    for k in (0, 4, 8, 16, 32, 64):
        a = one(k, 0, usec)
        b = one(k, k, usec) if k else a
        rec.append({"k": k, "ms_all_cus": round(a, 3), "ms_reserved": round(b, 3)})
        print(f"| {k} | {a:.3f} | {b:.3f} |", flush=True)
    print("\n```json\n" + json.dumps({"usec": usec, "rows": rec}) + "\n```")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
