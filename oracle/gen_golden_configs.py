"""TEST INFRASTRUCTURE ONLY -- configs and seeded-input helpers shared by the golden generator
(oracle/gen_golden.py, needs the reference) and the tests (which do not)."""
import torch

import diffnorm_oracle as O

TINY_EPS = O.EpsConfig(dim=64, latent_dim=16, depth=2, heads=4, dim_head=16, wavenet_layers=3, wavenet_stacks=2)
# chain config: the reference's LatentDiscreteModel always builds Model(dim, z) with default depth
CHAIN_EPS = O.EpsConfig(dim=64, latent_dim=8)
CHAIN_VAE = O.VaeConfig(dim=192, latent_dim=32)  # mults [4,3] -> z = 8
# conditional variant (use_cond=True): prompt dim 48, 8 resampled latents
TINY_EPS_COND = O.EpsConfig(dim=64, latent_dim=16, depth=2, heads=4, dim_head=16, wavenet_layers=3, wavenet_stacks=2, dim_prompt=48,
                            num_latents_m=8, resampler_depth=2)
FULL_EPS = O.EpsConfig()
FULL_VAE = O.VaeConfig()


def seeded(shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=dtype)


def ragged_lengths(B, T, seed, lo=None):
    g = torch.Generator().manual_seed(seed)
    lo = lo if lo is not None else max(1, T // 2)
    lens = torch.randint(lo, T + 1, (B,), generator=g)
    lens[0] = T  # one full-length utterance
    return lens
