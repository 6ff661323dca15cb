#!/usr/bin/env python3
"""Times the VAE ends of the path (BASELINE config 1 shape and the benchmark shape): encode_params and decode through the C ABI."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffnorm_amd import engine, synthetic

dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = synthetic.random_vae_state_dict(seed=0) if hasattr(synthetic, "random_vae_state_dict") else None
if sd is None:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import diffnorm_oracle as O
    sd = O.make_vae_state_dict(O.VaeConfig(), "bench")
ve = engine.VaeEngine(sd, dim=768, latent_dim=128, dtype=dtype, device=dev)
for B, T in ((64, 128), (32, 512)):
    feat = torch.randn(B, T, 768, device=dev)
    lens = torch.full((B,), T)
    z = torch.randn(B, T, 128, device=dev)
    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    te = timeit(lambda: ve.encode_params(feat))
    td = timeit(lambda: ve.decode(z, lens))
    fe = 4849664.0 * B * T
    fd = (272271360.0 + 18432.0 * T) * B * T
    print(f"VAE {dtype} [{B},{T}]: encode {te:.3f} ms ({fe/te/1e9:.0f} TFLOP/s)  decode {td:.3f} ms ({fd/td/1e9:.0f} TFLOP/s)  {B*T/(te+td)*1e3:.0f} frames/s")
