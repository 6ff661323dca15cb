import torch.nn as nn

MODEL_REGISTRY = {}
MODEL_DATACLASS_REGISTRY = {}
ARCH_MODEL_REGISTRY = {}
ARCH_MODEL_NAME_REGISTRY = {}
ARCH_MODEL_INV_REGISTRY = {}
ARCH_CONFIG_REGISTRY = {}


class BaseFairseqModel(nn.Module):
    @classmethod
    def add_args(cls, parser):
        pass

    @classmethod
    def build_model(cls, args, task):
        raise NotImplementedError

    def max_positions(self):
        return None


class FairseqEncoder(nn.Module):
    def __init__(self, dictionary=None):
        super().__init__()
        self.dictionary = dictionary


class FairseqEncoderModel(BaseFairseqModel):
    def __init__(self, encoder):
        super().__init__()
        self.encoder = encoder


def register_model(name, dataclass=None):
    def register_model_cls(cls):
        if name in MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model ({})".format(name))
        if not issubclass(cls, BaseFairseqModel):
            raise ValueError("Model ({}: {}) must extend BaseFairseqModel".format(name, cls.__name__))
        MODEL_REGISTRY[name] = cls
        return cls

    return register_model_cls


def register_model_architecture(model_name, arch_name):
    def register_model_arch_fn(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError("Cannot register model architecture for unknown model type ({})".format(model_name))
        if arch_name in ARCH_MODEL_REGISTRY:
            raise ValueError("Cannot register duplicate model architecture ({})".format(arch_name))
        if not callable(fn):
            raise ValueError("Model architecture must be callable ({})".format(arch_name))
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_MODEL_NAME_REGISTRY[arch_name] = model_name
        ARCH_MODEL_INV_REGISTRY.setdefault(model_name, []).append(arch_name)
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn

    return register_model_arch_fn
