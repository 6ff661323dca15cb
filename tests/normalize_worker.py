"""One process of tests/test_hip_exchange.py::test_normalize_is_the_same_on_one_rank_and_on_two: normalize() over the REAL
LatentDiscreteModel.ddim_sample on cuda:0 with the RECIPE-sized engines, batches large enough to land on the 256-row tiles
(16 utterances x 512 frames) next to a short last batch (3 utterances).  Run directly (world 1) or through torch.distributed.run
(two ranks over gloo sharing cuda:0; nothing that touched the GPU is re-exec'ed).  Writes <out>/lines_rank<r>.txt and the
reconstruction of every batch this rank produced to <out>/recon_rank<r>.npz."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    out_dir, dtype = sys.argv[1], sys.argv[2]
    group = None
    if "RANK" in os.environ:
        import torch.distributed as dist

        dist.init_process_group(backend="gloo")
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from diffnorm_amd import normalize as N
    from diffnorm_amd import synthetic
    from diffnorm_amd.latent_module import LatentDiscreteModel, SpeechVAEEncoderDecoder

    cfg = synthetic.eps_config()
    vae = SpeechVAEEncoderDecoder(dim=768, latent_dim=128, dtype=dtype)
    vae.load_state_dict(synthetic.random_vae_state_dict(768, 128, seed=1), strict=True)
    ldm = LatentDiscreteModel(types.SimpleNamespace(encoder=vae), 512, 128, timesteps=1000, dtype=dtype)
    ldm.model.load_state_dict(dict(synthetic.random_eps_state_dict(cfg, seed=0), **{"pos_embed._float_tensor": torch.zeros(1)}), strict=True)
    ldm = ldm.to(dev).eval()

    rng = np.random.RandomState(11)
    utts = []
    for i in range(16 + 16 + 3):  # batches of 16: two that route to the 256-row tiles (M = 8192) and a short one (M <= 1536)
        n = 512 if i % 16 == 0 else int(rng.randint(300, 513))
        units = (rng.randint(0, 500, size=n) * 2 + np.arange(n) % 2).tolist()  # neighbours differ: frame-level == de-duplicated
        utts.append(N.Utterance(f"utt{i}", f"src{i}.wav", 100 + i, torch.from_numpy(rng.randn(n, 768).astype(np.float32)), units, units))
    recons = {}

    def sample(feat, input_mask, cond_scale, ref_units, start_step):
        key = int(feat.shape[0]) * 100000 + int(input_mask.sum())
        g = torch.Generator().manual_seed(900 + key % 1000)
        post = torch.randn(feat.shape[0], feat.shape[1], 128, generator=g)
        start = torch.randn(feat.shape[0], feat.shape[1], 128, generator=g)
        pred, m, t, recon = ldm.ddim_sample(feat, input_mask=input_mask, cond_scale=cond_scale, ref_units=ref_units, start_step=start_step,
                                            post_noise=post, start_noise=start)
        recons[f"b{key}"] = recon.cpu().numpy()
        return pred, m, t, recon

    lines = N.normalize(sample, utts, start_step=4, batch_size=16, device=dev, group=group)
    with open(os.path.join(out_dir, f"lines_rank{rank}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    np.savez(os.path.join(out_dir, f"recon_rank{rank}.npz"), **recons)
    if "RANK" in os.environ:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback

        with open(os.path.join(sys.argv[1], f"error_rank{os.environ.get('RANK', '0')}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
