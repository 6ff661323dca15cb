"""fairseq registry access with a local fallback.

With fairseq importable, `register_model` / `register_model_architecture` / `register_task` /
`register_criterion` are fairseq's own decorators (reference fairseq/models/__init__.py:109-207,
fairseq/tasks/__init__.py:48-101, fairseq/registry.py:62-100); names that the DiffNorm fork already
registers are *replaced* so the plugin can be loaded with --user-dir inside the fork.  Without fairseq
(this image) the same decorator contract is provided locally: duplicate names raise ValueError, models must
subclass the base model class, architectures map to a config function.
"""
try:
    import fairseq.models as _fm
    import fairseq.tasks as _ft
    from fairseq.criterions import CRITERION_REGISTRY as _CRIT, FairseqCriterion, register_criterion as _reg_crit
    from fairseq.models import BaseFairseqModel, FairseqEncoder, FairseqEncoderModel
    from fairseq.tasks import FairseqTask

    HAVE_FAIRSEQ = True

    def _closure_sets(fn):
        """The `set` objects a decorator factory closes over (fairseq/registry.py:62-100 keeps the criterion registry's
        class-name set in a closure cell of `register_x`)."""
        return [c.cell_contents for c in (getattr(fn, "__closure__", None) or ()) if isinstance(c.cell_contents, set)]

    def _forget_class_name(old_cls, *name_sets):
        if old_cls is not None:
            for s in name_sets:
                s.discard(old_cls.__name__)

    def register_model(name):
        """fairseq's decorator, after dropping what the fork registered under `name` (fairseq/models/__init__.py:109-153
        rejects a duplicate model name)."""
        for t in (_fm.MODEL_REGISTRY, _fm.MODEL_DATACLASS_REGISTRY):
            t.pop(name, None)
        return _fm.register_model(name)

    def register_model_architecture(model_name, arch_name):
        """(fairseq/models/__init__.py:156-205: duplicate architecture names are rejected; the inverse table is a list)."""
        for t in (_fm.ARCH_MODEL_REGISTRY, _fm.ARCH_MODEL_NAME_REGISTRY, _fm.ARCH_CONFIG_REGISTRY):
            t.pop(arch_name, None)
        inv = _fm.ARCH_MODEL_INV_REGISTRY.get(model_name, [])
        while arch_name in inv:
            inv.remove(arch_name)
        return _fm.register_model_architecture(model_name, arch_name)

    def register_task(name):
        """(fairseq/tasks/__init__.py:48-101: rejects a duplicate task name AND a duplicate class name -- the plugin's task
        classes carry the fork's class names, so the old class's name leaves TASK_CLASS_NAMES together with its entry)."""
        old = _ft.TASK_REGISTRY.pop(name, None)
        _ft.TASK_DATACLASS_REGISTRY.pop(name, None)
        _forget_class_name(old, _ft.TASK_CLASS_NAMES)

        def deco(cls):
            _ft.TASK_CLASS_NAMES.discard(cls.__name__)  # a same-named class registered under another task name
            return _ft.register_task(name)(cls)
        return deco

    def register_criterion(name):
        """(fairseq/registry.py:62-100: duplicate name and duplicate class name are both rejected)."""
        old = _CRIT.pop(name, None)
        sets = _closure_sets(_reg_crit)
        _forget_class_name(old, *sets)

        def deco(cls):
            for s in sets:
                s.discard(cls.__name__)
            return _reg_crit(name)(cls)
        return deco

    MODEL_REGISTRY, ARCH_MODEL_REGISTRY, ARCH_CONFIG_REGISTRY = _fm.MODEL_REGISTRY, _fm.ARCH_MODEL_REGISTRY, _fm.ARCH_CONFIG_REGISTRY
    TASK_REGISTRY, CRITERION_REGISTRY = _ft.TASK_REGISTRY, _CRIT

except ImportError:
    import torch.nn as nn

    HAVE_FAIRSEQ = False
    MODEL_REGISTRY, ARCH_MODEL_REGISTRY, ARCH_CONFIG_REGISTRY = {}, {}, {}
    TASK_REGISTRY, CRITERION_REGISTRY = {}, {}

    class BaseFairseqModel(nn.Module):
        @classmethod
        def add_args(cls, parser):
            pass

        @classmethod
        def build_model(cls, args, task):
            raise NotImplementedError

        def get_targets(self, sample, net_output):
            return sample["target"]

        def max_positions(self):
            return None

    class FairseqEncoder(nn.Module):
        def __init__(self, dictionary=None):
            super().__init__()
            self.dictionary = dictionary

        def max_positions(self):
            return 1e6

    class FairseqEncoderModel(BaseFairseqModel):
        """reference fairseq/models/fairseq_model.py:534-574."""

        def __init__(self, encoder):
            super().__init__()
            self.encoder = encoder

        def max_positions(self):
            return self.encoder.max_positions()

    class FairseqTask:
        def __init__(self, cfg=None):
            self.cfg = cfg
            self.datasets = {}

        @classmethod
        def add_args(cls, parser):
            pass

        def build_criterion(self, args):
            return CRITERION_REGISTRY[args.criterion](self)

    class FairseqCriterion(nn.Module):
        def __init__(self, task):
            super().__init__()
            self.task = task
            tgt = getattr(task, "target_dictionary", None)
            self.padding_idx = tgt.pad() if tgt is not None else -100

    def register_model(name):
        def deco(cls):
            if name in MODEL_REGISTRY:
                raise ValueError("Cannot register duplicate model ({})".format(name))
            if not issubclass(cls, BaseFairseqModel):
                raise ValueError("Model ({}: {}) must extend BaseFairseqModel".format(name, cls.__name__))
            MODEL_REGISTRY[name] = cls
            return cls
        return deco

    def register_model_architecture(model_name, arch_name):
        def deco(fn):
            if model_name not in MODEL_REGISTRY:
                raise ValueError("Cannot register model architecture for unknown model type ({})".format(model_name))
            if arch_name in ARCH_MODEL_REGISTRY:
                raise ValueError("Cannot register duplicate model architecture ({})".format(arch_name))
            if not callable(fn):
                raise ValueError("Model architecture must be callable ({})".format(arch_name))
            ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
            ARCH_CONFIG_REGISTRY[arch_name] = fn
            return fn
        return deco

    def register_task(name):
        def deco(cls):
            if name in TASK_REGISTRY:
                raise ValueError("Cannot register duplicate task ({})".format(name))
            if not issubclass(cls, FairseqTask):
                raise ValueError("Task ({}: {}) must extend FairseqTask".format(name, cls.__name__))
            TASK_REGISTRY[name] = cls
            return cls
        return deco

    def register_criterion(name):
        def deco(cls):
            if name in CRITERION_REGISTRY:
                raise ValueError("Cannot register duplicate criterion ({})".format(name))
            CRITERION_REGISTRY[name] = cls
            return cls
        return deco
