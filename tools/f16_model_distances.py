"""CPU question, no GPU: would fp16 contraction operands (v_mfma_f32_16x16x32_f16 -- the bf16 rate on gfx950, 3 more mantissa
bits, but |x| <= 65504) put the engines inside north_star's 1e-2 max-abs budget?  Runs oracle/bf16_emulation.py's model of the
engines' arithmetic with its Rounder in "f16" and in "bf16" kind on the golden inputs and prints the distance of each to the fp32
golden outputs, plus the overflow / subnormal counters of every rounding point.

    python tools/f16_model_distances.py > profiles/r04_f16_model_distances.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bf16_emulation as E  # noqa: E402
import diffnorm_oracle as O  # noqa: E402
from gen_golden_configs import CHAIN_EPS, CHAIN_VAE, FULL_EPS, FULL_VAE, TINY_EPS, seeded  # noqa: E402

T_ = lambda a: torch.from_numpy(np.asarray(a))
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))


def maxerr(a, b):
    return (a.double() - b.double()).abs().max().item()


def mse(a, b):
    return ((a.double() - b.double()) ** 2).mean().item()


def line(name, ref, outs):
    print(f"{name}: " + "   ".join(f"{k}: max-abs {maxerr(v, ref):.3e} mse {mse(v, ref):.3e}" for k, v in outs.items())
          + f"   (ref rms {ref.double().pow(2).mean().sqrt().item():.3f})")


def counters(R):
    bad = {p: s for p, s in R.stats.items()}
    return "  f16 rounding points [elements, overflow, subnormal, flushed, max|x|]: " + "; ".join(
        f"{p} {s[0]}/{s[1]}/{s[2]}/{s[3]}/{s[4]:.3g}" for p, s in sorted(bad.items()))


def both(fn):
    outs = {}
    Rf = None
    for kind in ("bf16", "f16"):
        R = E.Rounder(kind=kind)
        with torch.no_grad():
            outs[kind] = fn(R)
        if kind == "f16":
            Rf = R
    return outs, Rf


def main():
    torch.set_num_threads(8)
    # tiny eps-predictor, t in {3, 500, 999}
    g = G("eps_tiny")
    sd = O.make_eps_state_dict(TINY_EPS, "tiny")
    x, t, lens = T_(g["x"]), T_(g["t"]), T_(g["lens"])
    mask = O.lengths_to_mask(lens, x.shape[1])
    outs, R = both(lambda R: E.eps_forward(sd, TINY_EPS, x, t, mask, R=R))
    ref = T_(g["eps"])
    for b in range(x.shape[0]):
        m = mask[b]
        line(f"tiny t={int(t[b])}", ref[b][m], {k: v[b][m] for k, v in outs.items()})
    print(counters(R))

    # BASELINE config 2, full size, t = 500
    g = G("eps_full_cfg2")
    sd = O.make_eps_state_dict(FULL_EPS, "full")
    x = seeded((8, 256, 128), 0)
    lens, t = T_(g["lens"]), T_(g["t"])
    mask = O.lengths_to_mask(lens, 256)
    outs, R = both(lambda R: E.eps_forward(sd, FULL_EPS, x, t, mask, R=R))
    ref = T_(g["eps"])
    line("cfg2 [8,256] t=500", ref[mask], {k: v[mask] for k, v in outs.items()})
    print(counters(R))
    for tt in (3, 999):  # no fp32 golden at these t for the full size: against the fp32 oracle
        tv = torch.full_like(t, tt)
        with torch.no_grad():
            ref_t = O.eps_forward(sd, FULL_EPS, x, tv, mask)
        outs, R = both(lambda R: E.eps_forward(sd, FULL_EPS, x, tv, mask, R=R))
        line(f"full size [8,256] t={tt} (vs the fp32 oracle)", ref_t[mask], {k: v[mask] for k, v in outs.items()})
        print(counters(R))

    # small VAE
    vsd = O.make_vae_state_dict(CHAIN_VAE, "chain")
    feat = seeded((3, 48, CHAIN_VAE.dim), 31)
    lens = torch.tensor([48, 29, 40])
    mask = O.lengths_to_mask(lens, 48)
    p_ref = O.vae_encode_params(vsd, CHAIN_VAE, feat)
    outs, R = both(lambda R: E.vae_encode_params(vsd, CHAIN_VAE, feat, R=R))
    line("small VAE posterior parameters", p_ref, outs)
    z = O.posterior_sample(p_ref, seeded((3, 48, CHAIN_VAE.z), 5))
    r_ref, l_ref = O.vae_decode(vsd, CHAIN_VAE, z, mask)
    outs, R2 = both(lambda R: E.vae_decode(vsd, CHAIN_VAE, z, mask, R=R))
    line("small VAE recon", r_ref[mask], {k: v[0][mask] for k, v in outs.items()})
    line("small VAE logits", l_ref[mask], {k: v[1][mask] for k, v in outs.items()})
    print(counters(R2))

    # BASELINE config 1 (4 of the 64 utterances)
    g = G("vae_full_cfg1")
    sd = O.make_vae_state_dict(FULL_VAE, "full")
    feat = seeded((64, 128, 768), 0)[:4]
    lens = T_(g["lens"])[:4]
    mask = O.lengths_to_mask(lens, 128)
    with torch.no_grad():
        p_ref = O.vae_encode_params(sd, FULL_VAE, feat)
    outs, R = both(lambda R: E.vae_encode_params(sd, FULL_VAE, feat, R=R))
    line("cfg1 VAE posterior parameters", p_ref, outs)
    print(counters(R))
    z = O.posterior_sample(p_ref, seeded((64, 128, 128), 3)[:4])
    with torch.no_grad():
        r_ref, l_ref = O.vae_decode(sd, FULL_VAE, z, mask)
    outs, R = both(lambda R: E.vae_decode(sd, FULL_VAE, z, mask, R=R))
    line("cfg1 VAE recon", r_ref[mask], {k: v[0][mask] for k, v in outs.items()})
    line("cfg1 VAE logits", l_ref[mask], {k: v[1][mask] for k, v in outs.items()})
    print(counters(R))

    # DDIM chains on the small pair, start_step 5 and 50 (golden: the reference's recon)
    g = G("chain_small")
    esd, vsd = O.make_eps_state_dict(CHAIN_EPS, "chain"), O.make_vae_state_dict(CHAIN_VAE, "chain")
    B, Tn = 3, 48
    feat = seeded((B, Tn, CHAIN_VAE.dim), 31)
    lens = T_(g["lens"])
    mask = O.lengths_to_mask(lens, Tn)
    tab = O.ddpm_tables(200)
    for start in (5, 50):
        if f"s{start}_post_noise" not in g:
            continue

        def chain(R):
            p = E.vae_encode_params(vsd, CHAIN_VAE, feat, R=R)
            z = O.posterior_sample(p, T_(g[f"s{start}_post_noise"]))
            ts = torch.full((B,), start, dtype=torch.long)
            xe = tab.at("sqrt_alphas_cumprod", ts, 3) * z + tab.at("sqrt_one_minus_alphas_cumprod", ts, 3) * T_(g[f"s{start}_start_noise"])
            for tt in range(start - 1, 0, -1):
                t = torch.full((B,), tt, dtype=torch.long)
                xe = O.ddim_update(tab, xe, E.eps_forward(esd, CHAIN_EPS, xe, t, mask, R=R), t)
            return E.vae_decode(vsd, CHAIN_VAE, xe, mask, R=R)[0]

        outs, R = both(chain)
        ref = T_(g[f"s{start}_recon"])
        line(f"chain start={start} recon (vs the reference)", ref[mask], {k: v[mask] for k, v in outs.items()})
        print(counters(R))


if __name__ == "__main__":
    main()
