"""`--arch speech_vae_decoder`: the speech VAE (reference speech_vae_decoder.py:25-136) on the HIP engine."""
import torch

from ...latent_module import SpeechVAEEncoderDecoder, lengths_to_mask
from ..registry import FairseqEncoderModel, register_model, register_model_architecture
from .common_args import add_inherited_args, apply_arch_defaults, is_training_run


@register_model("speech_vae_decoder")
class SpeechVAEDecoder(FairseqEncoderModel):
    def __init__(self, args, encoder):
        super().__init__(encoder)
        self.args = args

    def forward(self, target_feature, target_unit, **model_kwargs):
        """-> (mse_loss, lm_logits [B,T,1004], kl_loss) (reference :35-44)."""
        tgt_mask = lengths_to_mask(model_kwargs["tgt_lengths"], target_feature.shape[1])
        extra = {k: model_kwargs[k] for k in ("noise", "ntokens", "return_stats") if model_kwargs.get(k) is not None}
        return self.encoder(target_feature, target_unit, tgt_mask, **extra)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output[0]
        return torch.log_softmax(logits, dim=-1) if log_probs else torch.softmax(logits, dim=-1)

    @classmethod
    def build_model(cls, args, task):
        encoder = SpeechVAEEncoderDecoder(dim=getattr(args, "feature_dim", 768), latent_dim=args.latent_dim,
                                          dtype=getattr(args, "hip_dtype", "bf16"))
        encoder.train_on_move = is_training_run(args)  # the training engine (one flat parameter) comes up with model.to(device)
        return cls(args, encoder)

    @staticmethod
    def add_args(parser):
        add_inherited_args(parser)
        parser.add_argument("--latent_dim", type=int, default=16)
        parser.add_argument("--hip-dtype", default="bf16", choices=["bf16", "f16", "bf16x3", "f32"], help="MFMA arithmetic of the HIP engine (f16: IEEE-half operands, the 2-byte mode inside the 1e-2 budget, inference only; bf16: fastest, also the fast training mode; bf16x3: split-operand bf16, fp32-class results, inference only; f32: exact)")

    def max_positions(self):
        return self.encoder.max_positions()


@register_model_architecture("speech_vae_decoder", "speech_vae_decoder")
def base_architecture(args):
    apply_arch_defaults(args)
