"""`--arch diff_discrete`: latent diffusion over the frozen VAE's latents (reference diff_discrete.py:25-172)."""
import types

import torch

from ...latent_module import LatentDiscreteModel, SpeechVAEEncoderDecoder, lengths_to_mask
from ..registry import HAVE_FAIRSEQ, FairseqEncoderModel, register_model, register_model_architecture
from .common_args import add_inherited_args, apply_arch_defaults, is_training_run


def _load_speech_decoder(args):
    """The reference loads the VAE with fairseq's checkpoint utils from --speech_decoder_ckpt (:70-78); without
    fairseq the plugin reads the same file's state["model"] (keys encoder.*) into its own VAE module."""
    dtype = getattr(args, "hip_dtype", "bf16")
    path = getattr(args, "speech_decoder_ckpt", None)
    if HAVE_FAIRSEQ and path:  # pragma: no cover
        import fairseq

        models, _, _ = fairseq.checkpoint_utils.load_model_ensemble_and_task([path])
        return models[0]
    vae = SpeechVAEEncoderDecoder(dim=getattr(args, "feature_dim", 768), latent_dim=args.latent_dim, dtype=dtype)
    if path:
        state = torch.load(path, map_location="cpu", weights_only=False)
        sd = state["model"] if "model" in state else state
        vae.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, strict=True)
    return types.SimpleNamespace(encoder=vae)


@register_model("diff_discrete")
class DiffDiscreteModel(FairseqEncoderModel):
    def __init__(self, args, encoder):
        super().__init__(encoder)
        self.args = args

    def forward(self, target_feature, target_unit, **model_kwargs):
        """-> {total_loss, nll_loss, recon_mse_loss, noise_loss, acc} (reference :42-55)."""
        T = target_feature.shape[1]
        src = model_kwargs.get("src_feature")
        src_mask = lengths_to_mask(model_kwargs["src_lengths"]) if model_kwargs.get("src_lengths") is not None else None
        tgt_mask = lengths_to_mask(model_kwargs["tgt_lengths"], T)
        draws = {k: model_kwargs[k] for k in ("times", "post_noise", "jitter_noise", "true_noise") if model_kwargs.get(k) is not None}
        return self.encoder(target_feature, target_unit, src_feature=src, src_mask=src_mask, tgt_mask=tgt_mask,
                            unk_token=model_kwargs.get("unk_token"), **draws)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output[0]
        return torch.log_softmax(logits, dim=-1) if log_probs else torch.softmax(logits, dim=-1)

    @classmethod
    def build_model(cls, args, task):
        speech_decoder = _load_speech_decoder(args)
        speech_decoder.encoder.eval()
        for p in speech_decoder.encoder.parameters():
            p.requires_grad = False
        # upstream passes the --latent_dim flag as the denoiser's channel count (:84), which equals the VAE's latent width at
        # the recipe's feature dim 768; asking the VAE keeps other feature dims (tests) consistent
        vae = speech_decoder.encoder
        z = vae.latent_channels() if hasattr(vae, "latent_channels") else args.latent_dim
        encoder = LatentDiscreteModel(speech_decoder, getattr(args, "denoiser_dim", 512), z, timesteps=getattr(args, "diffusion_timesteps", 200),
                                      multitask=args.multitask, dtype=getattr(args, "hip_dtype", "bf16"))
        encoder.train_on_move = is_training_run(args)  # the training engine (one flat parameter) comes up with model.to(device)
        return cls(args, encoder)

    @staticmethod
    def add_args(parser):
        add_inherited_args(parser)
        parser.add_argument("--speech_decoder_ckpt", type=str, help="path to the speech decoder checkpoint")
        parser.add_argument("--use_cond", type=bool, default=False, help="use conditional diffusion")
        parser.add_argument("--latent_dim", type=int, default=16)
        parser.add_argument("--multitask", type=bool, default=False)
        parser.add_argument("--diffusion-timesteps", type=int, default=200,
                            help="schedule length (upstream hard-codes 200 at diff_discrete.py:84)")
        parser.add_argument("--hip-dtype", default="bf16", choices=["bf16", "f16", "bf16x3", "f32"], help="MFMA arithmetic of the HIP engine (f16: IEEE-half operands, the 2-byte mode inside the 1e-2 budget, inference only; bf16: fastest, also the fast training mode; bf16x3: split-operand bf16, fp32-class results, inference only; f32: exact)")

    def max_positions(self):
        return self.encoder.max_positions()


@register_model_architecture("diff_discrete", "diff_discrete")
def base_architecture(args):
    apply_arch_defaults(args)


@register_model_architecture("diff_discrete", "speech_diffusion")  # the name BASELINE.json uses for this arch
def speech_diffusion_alias(args):
    apply_arch_defaults(args)
