"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffnorm_amd/).

Host restatement of the attention-dropout mask of the HIP kernels (diffnorm_amd/csrc/common.h: dn_mix32 / dn_drop_row /
dn_drop_keep).  The reference draws its mask from torch's generator (nn.Dropout, latent_module.py:338), which no other
implementation can re-draw; what the tests pin instead is (a) Bernoulli(1 - p) statistics, (b) that forward and backward use the
same mask, (c) parity of outputs / gradients with the oracle when the oracle is handed this mask."""
import numpy as np
import torch


def dropout_keep_mask(B, heads, T, Tk, p, seed):
    """Host restatement of the kernels' dropout mask (csrc/common.h dn_drop_row / dn_drop_keep): bool [B, heads, T, Tk], True =
    kept.  Entry (b, h, i, j) depends only on (seed, b, h, i, j), which is what lets the backward re-derive it."""
    def mix(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16)
        x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
        x ^= x >> np.uint32(15)
        x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return x

    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        row = np.arange(B * heads * T, dtype=np.uint32).reshape(B, heads, T, 1)
        rh = mix(row ^ lo)
        key = np.arange(Tk, dtype=np.uint32).reshape(1, 1, 1, Tk)
        h = mix((rh + key * np.uint32(0x9E3779B9) + hi).astype(np.uint32))
    thr = np.uint32(min(np.float32(p) * np.float32(4294967296.0), np.float32(4294967040.0)))
    return torch.from_numpy(h >= thr)


def layer_keep(p, seed_lo, seed_hi):
    """keep(layer, B, heads, T, Tk) for oracle.attention_dropout: the training engines hash layer l with seed_hi + l."""
    def keep(layer, B, heads, T, Tk):
        return dropout_keep_mask(B, heads, T, Tk, p, (((seed_hi + layer) & 0xFFFFFFFF) << 32) | seed_lo)
    return keep
