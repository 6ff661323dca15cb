"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffnorm_amd/).

Leaf-file loader for the upstream reference's hot-path modules, used ONLY in the
build container (where /root/reference is mounted) to

  * validate the CPU restatement in ``oracle/diffnorm_oracle.py`` and
  * generate the golden vectors committed under ``tests/golden/`` (see
    ``oracle/gen_golden.py``).

``import fairseq`` as a whole fails here (omegaconf/hydra are absent), so the hot-path
files are loaded one by one under their real dotted names with empty package stubs
around them (recipe: SURVEY.md section 10).  Nothing is copied out of the reference;
the modules are executed where they lie.  The reference never travels to the GPU box,
so nothing under tests/ marked ``gpu``, ``bench.py`` or ``__graft_entry__.smoke`` may
call this module.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("DIFFNORM_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "fairseq/models/text_to_speech/latent_module.py"))


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_CACHE = {}


def load_reference():
    """Returns (latent_module, gaussian_diffusion_pkg) of the reference."""
    if "lm" in _CACHE:
        return _CACHE["lm"], _CACHE["gd"]
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REF)
    sys.dont_write_bytecode = True  # the reference mount is read-only
    import torch

    for n in ("torchaudio", "torchaudio.transforms", "torchaudio.functional", "sacrebleu"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)  # imported, never used on the path
    fs = _pkg("fairseq")
    fs.utils = _load("fairseq.utils", "fairseq/utils.py")
    mods = _pkg("fairseq.modules")
    _load("fairseq.modules.learned_positional_embedding", "fairseq/modules/learned_positional_embedding.py")
    _load("fairseq.modules.sinusoidal_positional_embedding", "fairseq/modules/sinusoidal_positional_embedding.py")
    mods.PositionalEmbedding = _load(
        "fairseq.modules.positional_embedding", "fairseq/modules/positional_embedding.py"
    ).PositionalEmbedding
    _pkg("fairseq.models").FairseqEncoder = _load(
        "fairseq.models.fairseq_encoder", "fairseq/models/fairseq_encoder.py"
    ).FairseqEncoder
    _pkg("fairseq.criterions")
    ls = types.ModuleType("fairseq.criterions.label_smoothed_cross_entropy")

    # the real file's module imports drag omegaconf, but the function itself needs nothing: compile the reference's OWN
    # label_smoothed_nll_loss (fairseq/criterions/label_smoothed_cross_entropy.py:34-51) from its source text, so the NLL term of
    # the training fixtures is the reference's code, not a restatement
    import ast

    path = os.path.join(REF, "fairseq/criterions/label_smoothed_cross_entropy.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "label_smoothed_nll_loss")
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    label_smoothed_nll_loss = ns["label_smoothed_nll_loss"]
    ls.label_smoothed_nll_loss = label_smoothed_nll_loss
    sys.modules[ls.__name__] = ls
    _pkg("fairseq.models.text_to_speech")
    _load("fairseq.models.text_to_speech.distributions", "fairseq/models/text_to_speech/distributions.py")
    lm = _load("fairseq.models.text_to_speech.latent_module", "fairseq/models/text_to_speech/latent_module.py")
    p = os.path.join(REF, "fairseq/models/text_to_speech/diffusion")
    spec = importlib.util.spec_from_file_location(
        "refdiffusion", os.path.join(p, "__init__.py"), submodule_search_locations=[p]
    )
    gd = importlib.util.module_from_spec(spec)
    sys.modules["refdiffusion"] = gd
    spec.loader.exec_module(gd)
    _CACHE["lm"], _CACHE["gd"] = lm, gd
    return lm, gd


def load_reference_dataset():
    """Returns the reference's fairseq/data/audio/repr_to_repr_unit_dataset.py module and its real Dictionary class
    (SURVEY 8 f1).  The module's audio-pipeline imports (waveform/feature transforms, S2T dataset helpers) are not on
    the path and are replaced by empty stand-in names; the dataset logic itself runs as it lies."""
    if "ds" in _CACHE:
        return _CACHE["ds"], _CACHE["dict"]
    load_reference()
    for name, rel in (("fairseq.file_io", "fairseq/file_io.py"), ("fairseq.tokenizer", "fairseq/tokenizer.py"),
                      ("fairseq.file_chunker_utils", "fairseq/file_chunker_utils.py")):
        _load(name, rel)
    fd = _pkg("fairseq.data")
    fd.data_utils = _load("fairseq.data.data_utils", "fairseq/data/data_utils.py")
    fd.Dictionary = _load("fairseq.data.dictionary", "fairseq/data/dictionary.py").Dictionary
    fd.FairseqDataset = _load("fairseq.data.fairseq_dataset", "fairseq/data/fairseq_dataset.py").FairseqDataset
    fd.ConcatDataset = type("ConcatDataset", (), {})
    _pkg("fairseq.data.audio")

    class _NoTransform:
        @classmethod
        def from_config_dict(cls, cfg):
            return None

    def stub(modname, **names):
        m = types.ModuleType(modname)
        for k, v in names.items():
            setattr(m, k, v)
        sys.modules[modname] = m

    blank = lambda n: type(n, (), {})
    stub("fairseq.data.audio.audio_utils", get_features_or_waveform=None)
    stub("fairseq.data.audio.data_cfg", S2SDataConfig=blank("S2SDataConfig"))
    stub("fairseq.data.audio.speech_to_text_dataset", SpeechToTextDataset=blank("SpeechToTextDataset"),
         SpeechToTextDatasetCreator=blank("SpeechToTextDatasetCreator"), TextTargetMultitaskData=blank("TextTargetMultitaskData"),
         _collate_frames=None, _is_int_or_np_int=None)
    stub("fairseq.data.audio.feature_transforms", CompositeAudioFeatureTransform=_NoTransform)
    stub("fairseq.data.audio.waveform_transforms", CompositeAudioWaveformTransform=_NoTransform)
    stub("fairseq.data.audio.dataset_transforms", CompositeAudioDatasetTransform=_NoTransform)
    stub("fairseq.data.audio.speech_to_speech_dataset", SpeechToSpeechDataset=blank("SpeechToSpeechDataset"))
    ds = _load("fairseq.data.audio.repr_to_repr_unit_dataset", "fairseq/data/audio/repr_to_repr_unit_dataset.py")
    _CACHE["ds"], _CACHE["dict"] = ds, fd.Dictionary
    return ds, fd.Dictionary


def load_reference_optim():
    """Returns the reference's (Adam class, clip_grad_norm_, InverseSquareRootSchedule class) -- fairseq/optim/adam.py:97-239,
    fairseq/utils.py:347-397, fairseq/optim/lr_scheduler/inverse_square_root_schedule.py:31-85.  The registry /
    dataclass / omegaconf plumbing those files import is not on the path and is replaced by inert stand-ins; the optimizer
    arithmetic, the clipping and the schedule run as they lie."""
    if "optim" in _CACHE:
        return _CACHE["optim"]
    load_reference()
    import dataclasses

    def stub(modname, **names):
        m = types.ModuleType(modname)
        m.__path__ = []
        for k, v in names.items():
            setattr(m, k, v)
        sys.modules[modname] = m
        return m

    if "omegaconf" not in sys.modules:
        stub("omegaconf", II=lambda key: None, OmegaConf=type("OmegaConf", (), {}))

    @dataclasses.dataclass
    class FairseqDataclass:
        pass

    passthrough = lambda *a, **k: (lambda cls: cls)
    stub("fairseq.dataclass", FairseqDataclass=FairseqDataclass)
    stub("fairseq.optim", FairseqOptimizer=type("FairseqOptimizer", (), {}), register_optimizer=passthrough)
    stub("fairseq.optim.fused_adam", get_fused_adam_class=lambda: None)

    class FairseqLRScheduler:  # the base class's state only (fairseq/optim/lr_scheduler/fairseq_lr_scheduler.py)
        def __init__(self, cfg, optimizer):
            self.cfg, self.optimizer, self.best = cfg, optimizer, None

        def step(self, epoch, val_loss=None):
            pass

    stub("fairseq.optim.lr_scheduler", FairseqLRScheduler=FairseqLRScheduler, register_lr_scheduler=passthrough)
    adam = _load("fairseq.optim.adam", "fairseq/optim/adam.py")
    sched = _load("fairseq.optim.lr_scheduler.inverse_square_root_schedule",
                  "fairseq/optim/lr_scheduler/inverse_square_root_schedule.py")
    _CACHE["optim"] = (adam.Adam, sys.modules["fairseq.utils"].clip_grad_norm_, sched.InverseSquareRootSchedule)
    return _CACHE["optim"]


def load_reference_nar():
    """The REAL reference classes behind SURVEY 8 f4's loop, loaded where they lie: fairseq's TransformerDecoder stack
    (fairseq/models/transformer/transformer_decoder.py, fairseq/modules/{transformer_layer,multihead_attention,layer_norm,...}.py),
    the research model file research/TranSpeech/nar_transformer.py (TransformerUnitDecoder, NARS2UTTransformerModel.forward_decoder /
    initialize_output_tokens / regenerate_length_beam) and the research generator research/TranSpeech/iterative_refinement_generator.py.
    What the path does not touch is stubbed: the dataclass config machinery (omegaconf) -- `TransformerConfig.from_namespace` is
    replaced by a plain nested namespace with the attributes the decoder reads --, FSDP / checkpoint wrappers (identity), the
    speech encoder, CTC decoder and NAT base classes the model file merely imports or subclasses, and `ipdb`.
    -> namespace(nar=module, gen=module, TransformerDecoder=..., cfg_from_namespace=...)."""
    if "nar" in _CACHE:
        return _CACHE["nar"]
    load_reference()
    import torch
    import torch.nn as nn

    if "omegaconf" not in sys.modules:  # fairseq.utils.safe_getattr asks OmegaConf.is_config(obj) before a plain getattr: never a config here
        oc = types.ModuleType("omegaconf")
        oc.OmegaConf = type("OmegaConf", (), {"is_config": staticmethod(lambda obj: False)})
        sys.modules["omegaconf"] = oc
    fs = sys.modules["fairseq"]
    for name, rel in (("fairseq.incremental_decoding_utils", "fairseq/incremental_decoding_utils.py"),
                      ("fairseq.modules.fairseq_dropout", "fairseq/modules/fairseq_dropout.py"),
                      ("fairseq.modules.quant_noise", "fairseq/modules/quant_noise.py"),
                      ("fairseq.modules.layer_norm", "fairseq/modules/layer_norm.py"),
                      ("fairseq.modules.layer_drop", "fairseq/modules/layer_drop.py")):
        _load(name, rel)
    models = sys.modules["fairseq.models"]
    models.FairseqDecoder = _load("fairseq.models.fairseq_decoder", "fairseq/models/fairseq_decoder.py").FairseqDecoder
    models.FairseqIncrementalDecoder = _load("fairseq.models.fairseq_incremental_decoder",
                                             "fairseq/models/fairseq_incremental_decoder.py").FairseqIncrementalDecoder
    for n in ("FairseqEncoderModel", "FairseqEncoderDecoderModel", "FairseqLanguageModel"):
        setattr(models, n, type(n, (nn.Module,), {}))
    models.register_model = lambda name: (lambda cls: cls)
    models.register_model_architecture = lambda model_name=None, arch_name=None, *a, **k: (lambda fn: fn)
    mods = sys.modules["fairseq.modules"]
    mods.FairseqDropout = sys.modules["fairseq.modules.fairseq_dropout"].FairseqDropout
    mods.LayerNorm = sys.modules["fairseq.modules.layer_norm"].LayerNorm
    mods.LayerDropModuleList = sys.modules["fairseq.modules.layer_drop"].LayerDropModuleList
    mods.SinusoidalPositionalEmbedding = sys.modules["fairseq.modules.sinusoidal_positional_embedding"].SinusoidalPositionalEmbedding
    gelu_mod = _load("fairseq.modules.gelu", "fairseq/modules/gelu.py")  # fairseq.utils.get_activation_fn imports both names (relu is what runs)
    mods.gelu, mods.gelu_accurate = gelu_mod.gelu, gelu_mod.gelu_accurate
    mods.AdaptiveSoftmax = type("AdaptiveSoftmax", (nn.Module,), {})
    mods.BaseLayer = type("BaseLayer", (nn.Module,), {})
    mods.MultiheadAttention = _load("fairseq.modules.multihead_attention", "fairseq/modules/multihead_attention.py").MultiheadAttention
    ck = types.ModuleType("fairseq.modules.checkpoint_activations")
    ck.checkpoint_wrapper = lambda m, **k: m
    sys.modules[ck.__name__] = ck
    dist = types.ModuleType("fairseq.distributed")
    dist.fsdp_wrap = lambda m, **k: m
    sys.modules[dist.__name__] = dist
    fs.distributed = dist

    def cfg_from_namespace(args):
        """The attributes TransformerDecoderBase / TransformerDecoderLayerBase read (transformer_decoder.py, transformer_layer.py),
        as the dataclass's from_namespace would nest them."""
        if hasattr(args, "quant_noise") and hasattr(args, "decoder"):  # already nested (the dataclass's from_namespace is idempotent too)
            return args
        g = lambda k, d=None: getattr(args, k, d)
        side = lambda p: types.SimpleNamespace(embed_dim=g(p + "_embed_dim"), ffn_embed_dim=g(p + "_ffn_embed_dim"), layers=g(p + "_layers"),
                                               attention_heads=g(p + "_attention_heads"), normalize_before=g(p + "_normalize_before", False),
                                               learned_pos=g(p + "_learned_pos", False), layerdrop=g(p + "_layerdrop", 0.0),
                                               xformers_att_config=None, output_dim=g(p + "_output_dim", g(p + "_embed_dim")),
                                               input_dim=g(p + "_input_dim", g(p + "_embed_dim")))
        return types.SimpleNamespace(
            encoder=side("encoder"), decoder=side("decoder"), dropout=g("dropout", 0.0), attention_dropout=g("attention_dropout", 0.0),
            activation_dropout=g("activation_dropout", 0.0), relu_dropout=0.0, activation_fn=g("activation_fn", "relu"), export=False,
            quant_noise=types.SimpleNamespace(pq=0, pq_block_size=8, scalar=0), cross_self_attention=False, tie_adaptive_weights=False,
            tie_adaptive_proj=False, adaptive_softmax_cutoff=None, adaptive_softmax_factor=4, adaptive_softmax_dropout=0, adaptive_input=False,
            share_decoder_input_output_embed=g("share_decoder_input_output_embed", False), offload_activations=False,
            no_token_positional_embeddings=g("no_token_positional_embeddings", False), no_scale_embedding=g("no_scale_embedding", False),
            no_decoder_final_norm=False, min_params_to_wrap=10 ** 12, max_target_positions=g("max_target_positions", 1024),
            layernorm_embedding=g("layernorm_embedding", False), checkpoint_activations=False, base_layers=0)

    tr = types.ModuleType("fairseq.models.transformer")
    tr.__path__ = []
    tr.TransformerConfig = type("TransformerConfig", (), {"from_namespace": staticmethod(cfg_from_namespace)})
    sys.modules[tr.__name__] = tr
    mods.transformer_layer = _load("fairseq.modules.transformer_layer", "fairseq/modules/transformer_layer.py")
    dec = _load("fairseq.models.transformer.transformer_decoder", "fairseq/models/transformer/transformer_decoder.py")
    tr.TransformerDecoder, tr.TransformerDecoderBase, tr.Linear = dec.TransformerDecoder, dec.TransformerDecoderBase, dec.Linear
    tr.TransformerModelBase = type("TransformerModelBase", (nn.Module,), {})

    def Embedding(num_embeddings, embedding_dim, padding_idx):  # fairseq/models/transformer/transformer_base.py:175-179 (the module drags the dataclasses)
        m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
        nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
        if padding_idx is not None:
            nn.init.constant_(m.weight[padding_idx], 0)
        return m

    tr.Embedding = Embedding
    # what nar_transformer.py imports besides the decoder stack
    gen_f = _load("fairseq.iterative_refinement_generator", "fairseq/iterative_refinement_generator.py")
    fs.checkpoint_utils = types.ModuleType("fairseq.checkpoint_utils")
    sys.modules["fairseq.checkpoint_utils"] = fs.checkpoint_utils
    sys.modules["ipdb"] = types.ModuleType("ipdb")
    fd = _pkg("fairseq.data") if "fairseq.data" not in sys.modules else sys.modules["fairseq.data"]
    if "fairseq.data.data_utils" not in sys.modules:
        du = types.ModuleType("fairseq.data.data_utils")
        du.lengths_to_padding_mask = lambda lens: torch.arange(int(lens.max()))[None, :] >= lens[:, None]
        sys.modules[du.__name__] = du
        fd.data_utils = du
    s2t = types.ModuleType("fairseq.models.speech_to_text")
    s2t.__path__ = []
    sys.modules[s2t.__name__] = s2t
    # The speech encoder (round 4): S2TTransformerEncoder as it lies (fairseq/models/speech_to_text/s2t_transformer.py:295-420), compiled from
    # its class statement alone -- the module around it imports the hub interface, checkpoint utilities and the model registry,
    # none of which the forward pass touches -- over the real Conv1dSubsampler (modules/convolution.py, torch only), the real
    # TransformerEncoderLayer / LayerNorm / FairseqDropout / PositionalEmbedding loaded above and FairseqEncoder.
    import ast as _ast
    import math as _math

    _pkg("fairseq.models.speech_to_text.modules")
    conv = _load("fairseq.models.speech_to_text.modules.convolution", "fairseq/models/speech_to_text/modules/convolution.py")
    s2t_path = os.path.join(REF, "fairseq/models/speech_to_text/s2t_transformer.py")
    s2t_tree = _ast.parse(open(s2t_path).read())
    enc_cls = next(n for n in s2t_tree.body if isinstance(n, _ast.ClassDef) and n.name == "S2TTransformerEncoder")
    enc_ns = {"FairseqEncoder": models.FairseqEncoder, "FairseqDropout": mods.FairseqDropout, "math": _math, "nn": nn, "torch": torch,
              "Conv1dSubsampler": conv.Conv1dSubsampler, "Conv2dSubsampler": conv.Conv2dSubsampler, "PositionalEmbedding": mods.PositionalEmbedding,
              "TransformerEncoderLayer": None, "LayerNorm": mods.LayerNorm,
              "lengths_to_padding_mask": sys.modules["fairseq.data.data_utils"].lengths_to_padding_mask, "__name__": "fairseq.models.speech_to_text.s2t_transformer"}
    s2t._encoder_src = (enc_cls, s2t_path, enc_ns)  # finished below, once transformer_layer is loaded
    s2t.S2TTransformerEncoder = None
    _pkg("fairseq.models.speech_to_speech")
    _pkg("fairseq.models.speech_to_speech.modules")
    ctc = types.ModuleType("fairseq.models.speech_to_speech.modules.ctc_decoder")
    ctc.CTCDecoder = type("CTCDecoder", (nn.Module,), {})
    sys.modules[ctc.__name__] = ctc
    _load("fairseq.models.speech_to_speech.modules.stacked_embedding", "fairseq/models/speech_to_speech/modules/stacked_embedding.py")
    nat = types.ModuleType("fairseq.models.nat")
    nat.__path__ = []
    nat.NATransformerModel = type("NATransformerModel", (nn.Module,), {})
    nat.FairseqNATDecoder = type("FairseqNATDecoder", (nn.Module,), {})
    nat.ensemble_decoder = lambda fn: fn  # (a single model: the decorator's ensemble branch is never taken)
    sys.modules[nat.__name__] = nat

    def _mean_pooling(enc_feats, src_masks):  # fairseq/models/nat/nonautoregressive_transformer.py:20-34, compiled from its source
        raise RuntimeError("replaced below")

    import ast

    def compile_fn(rel, fn_name, ns):
        path = os.path.join(REF, rel)
        tree = ast.parse(open(path).read())
        fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == fn_name)
        exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
        return ns[fn_name]

    nrt = types.ModuleType("fairseq.models.nat.nonautoregressive_transformer")
    nrt._mean_pooling = compile_fn("fairseq/models/nat/nonautoregressive_transformer.py", "_mean_pooling", {"torch": torch})
    sys.modules[nrt.__name__] = nrt
    cm = types.ModuleType("fairseq.models.nat.cmlm_transformer")
    cm._skeptical_unmasking = compile_fn("fairseq/models/nat/cmlm_transformer.py", "_skeptical_unmasking",
                                         {"new_arange": sys.modules["fairseq.utils"].new_arange})
    sys.modules[cm.__name__] = cm
    enc_cls, s2t_path, enc_ns = s2t._encoder_src
    enc_ns["TransformerEncoderLayer"] = mods.transformer_layer.TransformerEncoderLayer
    exec(compile(_ast.Module(body=[enc_cls], type_ignores=[]), s2t_path, "exec"), enc_ns)
    s2t.S2TTransformerEncoder = enc_ns["S2TTransformerEncoder"]
    nar = _load("refnar.nar_transformer", "research/TranSpeech/nar_transformer.py")
    gen = _load("refnar.iterative_refinement_generator", "research/TranSpeech/iterative_refinement_generator.py")
    out = types.SimpleNamespace(nar=nar, gen=gen, gen_fairseq=gen_f, TransformerDecoder=dec.TransformerDecoder, cfg_from_namespace=cfg_from_namespace)
    _CACHE["nar"] = out
    return out
