"""One rank of tests/test_hip_exchange.py (started through torch.distributed.run; never imported by pytest): the REAL training
engines and trainers of diffnorm_amd/training.py on cuda:0 with GPU gradient buffers, two ranks exchanging over gloo (RCCL
refuses two ranks on one device; the code path -- ready event, side stream, async all-reduce per bucket, statistics all-reduce,
finish -- is the one RCCL runs).  Each rank trains on its own batch and writes what it ended with to <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_batch(rank, cfg_dim, z=None, timesteps=200):
    """The batch of `rank` (different shapes per rank: B = 2 / 3, T = 40 / 48) and, for the diffusion loss, its injected draws."""
    g = torch.Generator().manual_seed(1000 + rank)
    B, T = (2, 40) if rank == 0 else (3, 48)
    lens = torch.tensor([T, T - 13, T - 5][:B])
    mask = torch.arange(T).view(1, -1) < lens.view(-1, 1)
    feat = torch.randn(B, T, cfg_dim, generator=g) * mask.unsqueeze(-1)
    unit = torch.randint(4, 1004, (B, T), generator=g) * mask
    sample = {"reduce_target": feat, "reduce_target_unit": unit, "reduce_target_lengths": lens, "ntokens": int(lens.sum()), "nsentences": B}
    draws = None
    if z is not None:
        draws = {"times": torch.randint(1, timesteps, (B,), generator=g), "post_noise": torch.randn(B, T, z, generator=g),
                 "jitter_noise": torch.randn(B, T, z, generator=g), "true_noise": torch.randn(B, T, z, generator=g)}
    return sample, draws


def build(kind, dev, group_world):
    import types

    import diffnorm_oracle as O
    from diffnorm_amd import training
    from gen_golden_configs import CHAIN_EPS, CHAIN_VAE as CFG

    vsd = O.make_vae_state_dict(CFG, "train")
    if kind == "vae":
        eng = training.VaeTrainEngine(vsd, dim=CFG.dim, latent_dim=CFG.latent_dim, dtype="f32", device=dev, depth=CFG.depth, heads=CFG.heads,
                                      dim_head=CFG.dim_head, stacks=CFG.stacks, layers=CFG.layers)
        tr = training.VaeTrainer(eng, lr=1e-3, clip_norm=2.0, warmup_updates=4, warmup_init_lr=1e-4, bucket_mb=0.25, attn_dropout=0.0)
        return tr, CFG, None
    from diffnorm_amd.latent_module import LatentDiscreteModel, SpeechVAEEncoderDecoder

    vae = SpeechVAEEncoderDecoder(dim=CFG.dim, latent_dim=CFG.latent_dim, dtype="f32")
    vae.load_state_dict(vsd, strict=True)
    ldm = LatentDiscreteModel(types.SimpleNamespace(encoder=vae), CHAIN_EPS.dim, CFG.z, timesteps=200, dtype="f32")
    esd = O.make_eps_state_dict(CHAIN_EPS, "train")
    ldm.model.load_state_dict(dict(esd, **{"pos_embed._float_tensor": torch.zeros(1)}), strict=True)
    ldm.to(dev)
    tr = training.DiffusionTrainer(ldm, lr=1e-3, clip_norm=2.0, warmup_updates=4, warmup_init_lr=1e-4, bucket_mb=0.25, attn_dropout=0.0)
    return tr, CFG, CFG.z


def run(kind, tr, batches, n_updates):
    """batches: the micro-batches of ONE update (one per rank in the 2-rank run; both in the single-process run)."""
    out = {}
    for it in range(n_updates):
        samples = [b[0] for b in batches]  # batches: (sample, draws, rank the batch belongs to)
        noises = [("philox", 50 + b[2], it << 20) for b in batches] if kind == "vae" else [b[1] for b in batches]
        logged, norm = tr.train_step(samples, noises=noises)
        out[f"norm{it}"] = float(norm)
        out[f"logged{it}"] = logged.cpu().numpy()
        if it == 0:
            out["grad0"] = tr.engine.grads.cpu().numpy().copy()  # the summed (post-exchange) gradient of the first update
    out["master"] = tr.engine.master.cpu().numpy()
    return out


def main():
    kind, out_dir = sys.argv[1], sys.argv[2]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    tr, cfg, z = build(kind, dev, world)
    assert tr.reducer.world == 2 and tr.reducer.cuda and len(tr.reducer.buckets) >= 3, (tr.reducer.world, len(tr.reducer.buckets))
    sample, draws = make_batch(rank, cfg.dim, z)
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        res = run(kind, tr, [(sample, draws, rank)], 3)
        torch.cuda.synchronize()
    res["buckets"] = len(tr.reducer.buckets)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback

        with open(os.path.join(sys.argv[2], f"error_rank{os.environ.get('RANK', '0')}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
