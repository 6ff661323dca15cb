#!/usr/bin/env python3
"""Diagnostic: in-kernel shader clock (s_memtime / s_memrealtime) and K-loop cycles of the 256x352 tile kernel on the
FFN-conv shape.  Needs a stamped build:  touch diffnorm_amd/csrc/gemm.hip && make -C diffnorm_amd/csrc EXTRA=-DDN_FAT_STAMPS
(add -DDN_FAT_ABL=1|2|3 to drop the DMA / 7 of 8 MFMAs / both).  Rebuild without EXTRA afterwards."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing
dev = torch.device("cuda:0")
M, K, N, T = 16384, 1408, 1408, 512
a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
w = [(torch.randn(1408, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(1408, device=dev)
dbg = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
lib = _lib.load()
def run(flag, abl=0):
    p = _lib.GemmParams()
    p.n_terms, p.dtype, p.M, p.N, p.K, p.T, p.groups, p.epilogue = 3, _lib.DN_BF16, M, N, K, T, 1, _lib.EPI_BIAS
    for i in range(3):
        t = p.terms[i]; t.A, t.W, t.lda, t.shift = a.data_ptr(), w[i].data_ptr(), K, 2 - i
    p.bias = bias.data_ptr(); p.out, p.ldo, p.out_dtype = out.data_ptr(), N, _lib.DN_BF16
    p.pos_table = dbg.data_ptr() if flag else 0
    p.pad_ = (4 << 16) | ((1 << 20) if flag else 0) | abl
    _lib.check(lib.dn_conv_gemm(C.byref(p), torch.cuda.current_stream().cuda_stream), "gemm")
for _ in range(20): run(False)
for abl in (0, 0, 0):  # repeats
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(True, abl); e1.record(); torch.cuda.synchronize()
    d = dbg.view(256, 4).cpu().double()
    pro, loop, rt, r0 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    mhz = ((pro + loop) / rt * 100).mean()
    print(f"abl {abl} launch {e0.elapsed_time(e1)*1e3:.1f} us | prologue {pro.mean():.0f} cyc, K-loop {loop.mean():.0f} cyc (min {loop.min():.0f} max {loop.max():.0f}) = {loop.mean()/132:.0f} per K-tile | "
          f"WG real time {rt.mean()/100:.1f} us (min {rt.min()/100:.1f} max {rt.max()/100:.1f}) | clock ~{mhz:.0f} MHz | start skew {(r0.max()-r0.min())/100:.1f} us")
