"""CLI flags the reference's two models accept.  Upstream inherits them from an unused DiT-style decoder
(reference diff_transformer.py:190-314, called at speech_vae_decoder.py:70 / diff_discrete.py:90); none of
them reaches the hot path, they are accepted so existing command lines keep working."""

_INHERITED = [
    ("--conv-kernel-sizes", dict(type=str, metavar="N")),
    ("--conv-channels", dict(type=int, metavar="N")),
    ("--activation-fn", dict(type=str, default="relu")),
    ("--dropout", dict(type=float, metavar="D")),
    ("--attention-dropout", dict(type=float, metavar="D")),
    ("--activation-dropout", dict(type=float, metavar="D")),
    ("--relu-dropout", dict(type=float, metavar="D")),
    ("--encoder-embed-dim", dict(type=int, metavar="N")),
    ("--encoder-ffn-embed-dim", dict(type=int, metavar="N")),
    ("--encoder-layers", dict(type=int, metavar="N")),
    ("--encoder-attention-heads", dict(type=int, metavar="N")),
    ("--encoder-normalize-before", dict(action="store_true")),
    ("--decoder-embed-dim", dict(type=int, metavar="N")),
    ("--decoder-ffn-embed-dim", dict(type=int, metavar="N")),
    ("--decoder-layers", dict(type=int, metavar="N")),
    ("--decoder-attention-heads", dict(type=int, metavar="N")),
    ("--decoder-normalize-before", dict(action="store_true")),
    ("--share-decoder-input-output-embed", dict(action="store_true")),
    ("--layernorm-embedding", dict(action="store_true")),
    ("--no-scale-embedding", dict(action="store_true")),
    ("--load-pretrained-encoder-from", dict(type=str, metavar="STR")),
    ("--encoder-freezing-updates", dict(type=int, metavar="N")),
    ("--speaker-embed-dim", dict(type=int, metavar="N")),
]


def add_inherited_args(parser):
    for flag, kw in _INHERITED:
        parser.add_argument(flag, **kw)
    # the three flags both model files add after the inherited ones
    parser.add_argument("--input-feat-per-channel", default=80)
    parser.add_argument("--depthwise-conv-kernel-size", default=31)
    parser.add_argument("--input-channels", default=1)
    parser.add_argument("--attn-type", default=None)
    parser.add_argument("--pos-enc-type", default="abs")
    parser.add_argument("--classifier_guidance", type=float, default=1.0)


_ARCH_DEFAULTS = dict(
    attn_type=None, pos_enc_type="abs", classifier_guidance=1.0, encoder_freezing_updates=0, conv_kernel_sizes="5,5",
    conv_channels=1024, conv_version="s2t_transformer", encoder_embed_dim=512, encoder_ffn_embed_dim=2048, encoder_layers=12,
    encoder_attention_heads=8, encoder_normalize_before=True, no_scale_embedding=False, dropout=0.1, activation_fn="relu",
    speaker_embed_dim=256, decoder_layers=6, decoder_attention_heads=8, decoder_normalize_before=True,
    decoder_learned_pos=False, adaptive_softmax_cutoff=None, adaptive_softmax_dropout=0,
    share_decoder_input_output_embed=False, no_token_positional_embeddings=False, adaptive_input=False,
    decoder_layerdrop=0.0, quant_noise_pq=0, length_loss_factor=0.1,
)


def apply_arch_defaults(args):
    """Fills the attributes the reference's base_architecture functions set (speech_vae_decoder.py:101-136)."""
    for k, v in _ARCH_DEFAULTS.items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    for k, src in (("attention_dropout", "dropout"), ("activation_dropout", "dropout"), ("decoder_embed_dim", "encoder_embed_dim"),
                   ("decoder_ffn_embed_dim", "encoder_ffn_embed_dim")):
        if getattr(args, k, None) is None:
            setattr(args, k, getattr(args, src))
    for k, src in (("decoder_output_dim", "decoder_embed_dim"), ("decoder_input_dim", "decoder_embed_dim")):
        if getattr(args, k, None) is None:
            setattr(args, k, getattr(args, src))


def is_training_run(args) -> bool:
    """Is this namespace a training run's (fairseq_cli/train.py: it carries --optimizer / --lr; generation and
    diff_norm_synthesis namespaces do not)?  `--hip-train-engine` / `hip_train_engine=True` forces it.  A training run's model
    switches to the HIP training engine -- ONE flat parameter -- the moment it is moved to the GPU, i.e. before the reference
    trainer snapshots `model.parameters()` into its optimizer (fairseq/trainer.py:292)."""
    if getattr(args, "hip_train_engine", None) is not None:
        return bool(args.hip_train_engine)
    return getattr(args, "optimizer", None) is not None

