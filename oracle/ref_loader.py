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
