import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffnorm_amd import ops, _lib, packing
dev = "cuda:0"
torch.manual_seed(0)
M, T, K, N = 2048, 256, 512, 1536
a = (torch.randn(M, K)).to(dev, torch.bfloat16)
w = (torch.randn(N, K) * 0.05).to(dev, torch.bfloat16)
bias = torch.randn(N, device=dev) * 0.1
def run(tile, **kw):
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.conv_gemm([(kw.pop("A", a), kw.pop("W", w), 0)], out, T, N, bias=bias, tile=tile, **kw)
    return out
r3, r9 = run(3), run(9)
print("row-major bias:", torch.equal(r3, r9))
akb, wkb = packing.kblock(a.cpu()).to(dev), packing.kblock(w.cpu()).to(dev)
k3 = run(3, A=akb, W=wkb, a_kblocked=True, w_kblocked=True)
k9 = run(9, A=akb, W=wkb, a_kblocked=True, w_kblocked=True)
print("k-blocked vs row-major (3):", torch.equal(k3, r3), " tile 9 k-blocked:", torch.equal(k9, r3), (k9.float() - r3.float()).abs().max().item())
ssq = torch.rand(M, 8, device=dev) + 0.5
rb = torch.randn(N, device=dev) * 0.1
s3 = run(3, row_ssq=ssq, row_D=512, row_bias=rb, row_bias_shared=True)
s9 = run(9, row_ssq=ssq, row_D=512, row_bias=rb, row_bias_shared=True)
print("row_ssq consumer:", torch.equal(s3, s9), (s3.float() - s9.float()).abs().max().item())
# GEGLU
ip = 704
wg = (torch.randn(2 * ip, K) * 0.05).to(dev, torch.bfloat16)
bg = torch.randn(2 * ip, device=dev) * 0.1
def geglu(tile, **kw):
    out = torch.empty(M, ip, device=dev, dtype=torch.bfloat16)
    ops.conv_gemm([(kw.pop("A", a), kw.pop("W", wg), 0)], out, T, ip, bias=bg, epilogue=_lib.EPI_GEGLU, tile=tile, **kw)
    return out
g3, g9 = geglu(3), geglu(9)
print("geglu:", torch.equal(g3, g9), (g3.float() - g9.float()).abs().max().item())
g3s, g9s = geglu(3, row_ssq=ssq, row_D=512), geglu(9, row_ssq=ssq, row_D=512)
print("geglu + row_ssq:", torch.equal(g3s, g9s), (g3s.float() - g9s.float()).abs().max().item())
wgkb = packing.kblock(wg.cpu()).to(dev)
g9k = geglu(9, A=akb, W=wgkb, a_kblocked=True, w_kblocked=True, row_ssq=ssq, row_D=512)
print("geglu k-blocked + row_ssq (9) vs (3):", torch.equal(g9k, g3s), (g9k.float() - g3s.float()).abs().max().item())
# which one is right?  fp64 reference of the BIAS + row_ssq + row_bias case
ref = (a.double() @ w.double().t()) * (512 ** 0.5 / ssq.double().sum(-1, keepdim=True).sqrt()) + rb.double() + bias.double()
for name, t in (("tile 3", s3), ("tile 9", s9)):
    e = (t.double() - ref).abs()
    ulp = torch.maximum(ref.abs(), torch.tensor(1e-30, device=dev)).log2().floor().exp2() * 2.0 ** -8
    print(name, "max err", e.max().item(), "max err in bf16 ulps", (e / ulp).max().item(), "mismatches vs RNE(ref)", int((t != ref.float().to(torch.bfloat16)).sum()))
d = (s3 != s9)
print("differing elements", int(d.sum()), "of", d.numel(), "rows with differences", int(d.any(1).sum()), "cols", int(d.any(0).sum()))
idx = d.nonzero()[:5]
print(idx.tolist())
