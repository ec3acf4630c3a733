"""Reference point, not product: the same GEMM shapes through torch.nn.functional.linear (hipBLASLt / rocBLAS) on the same box,
to read the hand-written kernels' numbers (tools/gemm_bench.py) against the vendor library.  GPU only."""
import torch, os
os.environ.setdefault("TORCH_BLAS_PREFER_HIPBLASLT","1")
M=25216
shapes=[("qkv",2304,768),("proj",768,768),("fc1",3072,768),("fc2",768,3072),("dqkv",768,2304)]
for name,N,K in shapes:
    a=torch.randn(M,K,device="cuda").to(torch.bfloat16); w=(torch.randn(N,K,device="cuda")*K**-0.5).to(torch.bfloat16); b=torch.randn(N,device="cuda").to(torch.bfloat16)
    f=lambda: torch.nn.functional.linear(a,w,b)
    for _ in range(5): f()
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
    t=sorted(ts)[2]
    print(f"{name:5s} N={N} K={K}: torch.linear {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:7.1f} TF", flush=True)
