#!/usr/bin/env python3
"""Times the contraction shapes of one denoising step ([B=32,T=512]) through the C ABI (HIP events)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import _lib, ops, packing

dev = torch.device("cuda:0")
B, T = int(os.environ.get("DN_BENCH_B", "32")), 512
M = B * T
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def mk(rows, cols):
    return (torch.randn(rows, cols, device=dev) * 0.5).to(dt)


rows = []
TILE = int(os.environ.get("DN_BENCH_TILE", "0"))  # forced tile variant for every shape (0 = the library's own choice)
def shape(name, K, N, taps=1, groups=1, epi=_lib.EPI_BIAS, count=1, kb=False):
    Kp, Np = packing.padk(K), packing.padk(N)
    a = mk(M, Kp) if groups == 1 else (torch.randn(groups, M, Kp, device=dev) * 0.5).to(dt)
    nrows = packing.padn(N) if epi != _lib.EPI_GEGLU else 2 * Np
    w = [(torch.randn(*( (groups,) if groups > 1 else ()), nrows, Kp, device=dev) * 0.02).to(dt) for _ in range(taps)]
    out_dt = torch.float32 if epi == _lib.EPI_RESADD else dt
    out = torch.empty(*((groups,) if groups > 1 else ()), M, Np, device=dev, dtype=out_dt)
    bias = torch.zeros(*((groups,) if groups > 1 else ()), nrows, device=dev)
    kw = {}
    if epi == _lib.EPI_RESADD:
        kw["res"] = out
    if epi == _lib.EPI_FILM_GATE:
        kw["res"] = torch.zeros_like(out)
        kw["gamma_beta"] = torch.ones(1, groups * 2 * Np, device=dev)
        kw["gb_shared"] = True
        kw["gb_half"] = Np
        kw["shift_by_group"] = True
    if kb:  # K-blocked operands ([K/32][rows][32])
        a, w = packing.kblock(a), [packing.kblock(x) for x in w]
        kw.update(a_kblocked=True, w_kblocked=True)
    terms = [(a, w[j], taps - 1 - j) for j in range(taps)]
    sec = timeit(lambda: ops.conv_gemm(terms, out, T, Np, bias=bias, epilogue=epi, groups=groups, tile=TILE, **kw))
    flops = 2.0 * M * K * taps * N * groups * (2 if epi == _lib.EPI_GEGLU else 1)
    rows.append((name, count, sec * 1e6, flops / sec / 1e12, count * sec * 1e3))
    print(f"{name:28s} x{count:2d}  {sec*1e6:8.1f} us  {flops/sec/1e12:7.1f} TF/s   {count*sec*1e3:6.3f} ms/step", flush=True)

def ffn_kblocked(a_kb, w_kb):
    """The FFN causal conv with K-blocked operands ([K/32][rows][32]) on the 256x352 tile."""
    Kp = packing.padk(1365)
    a, w = mk(M, Kp), (torch.randn(3, packing.padn(1365), Kp, device=dev) * 0.02).to(dt)
    ab, wb = packing.kblock(a), packing.kblock(w)
    out, bias = torch.empty(M, Kp, device=dev, dtype=dt), torch.zeros(packing.padn(1365), device=dev)
    terms = [(ab if a_kb else a, (wb if w_kb else w)[j], 2 - j) for j in range(3)]
    sec = timeit(lambda: ops.conv_gemm(terms, out, T, Kp, bias=bias, a_kblocked=a_kb, w_kblocked=w_kb), iters=60)
    fl = 2.0 * M * 1365 * 3 * 1365
    print(f"ffn_conv A {'K-blocked' if a_kb else 'row-major'}, W {'K-blocked' if w_kb else 'row-major'}: {sec*1e6:8.1f} us  {fl/sec/1e12:7.1f} TF/s", flush=True)


only = sys.argv[2] if len(sys.argv) > 2 else None
if only == "ffn":  # as the engine runs it at this size: K-blocked operands on the 256x352 tile (PMC passes use this)
    ffn_kblocked(True, True)
    sys.exit(0)
if only == "ffn_rowmajor":
    shape("ffn_conv k3 1365->1365", 1365, 1365, taps=3, count=12)
    sys.exit(0)
if only == "shortk":  # the K = 512 shapes on the 256 x 256 tile (diagnostic builds: make EXTRA=-DDN_GEMM_ABL=<bits>)
    TILE = 3
    for rep_ in range(2):
        shape("ffn_in GEGLU 512->2x1365", 512, 1365, epi=_lib.EPI_GEGLU, count=12)
        shape("qkv 512->1536", 512, 1536, count=12)
        shape("wn_res 1x1 512 g8", 512, 512, groups=8, count=4)
    sys.exit(0)
if only == "kblock256":  # the 256 x 256 tile's shapes, row-major vs K-blocked, alternating
    TILE = 3
    for rep_ in range(3):
        for kb in (False, True):
            print("K-blocked" if kb else "row-major", flush=True)
            shape("wn_dilated k3 512 g8", 512, 512, taps=3, groups=8, epi=_lib.EPI_FILM_GATE, count=4, kb=kb)
            shape("wn_res 1x1 512 g8", 512, 512, groups=8, count=4, kb=kb)
            shape("ffn_in GEGLU 512->2x1365", 512, 1365, epi=_lib.EPI_GEGLU, count=12, kb=kb)
            shape("qkv 512->1536", 512, 1536, count=12, kb=kb)
    sys.exit(0)
if only == "sizes":  # how the causal-conv contraction scales with channel width on each tile variant (DN_BENCH_TILE)
    for width in (1024, 1365, 1408, 1536, 2048, 2731):
        shape(f"conv k3 {width}->{width}", width, width, taps=3)
    sys.exit(0)
if only == "kblock":
    for a_kb, w_kb in ((False, False), (True, False), (False, True), (True, True)) + ((False, False), (True, True)) * 4:
        ffn_kblocked(a_kb, w_kb)
    sys.exit(0)
shape("ffn_conv k3 1365->1365", 1365, 1365, taps=3, count=12)
shape("wn_dilated k3 512 g8", 512, 512, taps=3, groups=8, epi=_lib.EPI_FILM_GATE, count=4)
shape("wn_res 1x1 512 g8", 512, 512, groups=8, count=4)
shape("ffn_in GEGLU 512->2x1365", 512, 1365, epi=_lib.EPI_GEGLU, count=12)
shape("ffn_out 1365->512 resadd", 1365, 512, epi=_lib.EPI_RESADD, count=12)
shape("qkv 512->1536", 512, 1536, count=12)
shape("attn_out 512->512 resadd", 512, 512, epi=_lib.EPI_RESADD, count=12)
shape("wn_init k3 512", 512, 512, taps=3, count=1)
shape("linear 512->512", 512, 512, count=3)
print("total GEMM ms/step: %.3f" % sum(r[4] for r in rows))

# attention [B=32,T=512,H=8,dh=64]
hd = 512
qkv = (torch.randn(M, 3 * hd, device=dev) * 0.5).to(dt)
ao = torch.empty(M, hd, device=dev, dtype=dt)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
sec = timeit(lambda: ops.attention(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ao, B, T, 8, 64, lens, ldq=3 * hd, ldk=3 * hd, ldv=3 * hd))
fl = 4.0 * T * T * 64 * 8 * B
print(f"{'attention 8x64 T=512':28s} x12  {sec*1e6:8.1f} us  {fl/sec/1e12:7.1f} TF/s   {12*sec*1e3:6.3f} ms/step")
xr = torch.randn(M, 512, device=dev)
xn = torch.empty(M, 512, device=dev, dtype=dt)
gbv = torch.randn(1, 1024, device=dev)
sec = timeit(lambda: ops.rmsnorm(xr, xn, T, gamma_beta=gbv, gb_shared=True, gb_half=512))
print(f"{'rmsnorm 512':28s} x25  {sec*1e6:8.1f} us  {M*512*6/sec/1e12:7.2f} TB/s   {25*sec*1e3:6.3f} ms/step")
