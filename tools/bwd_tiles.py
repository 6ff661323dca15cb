import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import ops, packing
dev="cuda:0"
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for C, B in ((2048, 24), (2048, 30), (1365, 16)):
    T=512; M=B*T; Cp=packing.padk(C)
    dy=(torch.randn(M,Cp,device=dev)*0.5).bfloat16()
    wt=(torch.randn(3,packing.padn(C),Cp,device=dev)*0.02).bfloat16()
    dx=torch.empty(M,Cp,device=dev,dtype=torch.bfloat16)
    for tile in (0,1,2,3):
        t=timeit(lambda: ops.conv_gemm([(dy, wt[j], -(2-j)) for j in range(3)], dx, T, Cp, tile=tile))
        print(f"dX conv {C} M={M} tile {tile}: {t:8.1f} us", flush=True)
