TASK_DATACLASS_REGISTRY = {}
TASK_REGISTRY = {}
TASK_CLASS_NAMES = set()


class FairseqTask:
    def __init__(self, cfg=None, **kwargs):
        self.cfg = cfg
        self.datasets = {}

    @classmethod
    def add_args(cls, parser):
        pass


def register_task(name, dataclass=None):
    def register_task_cls(cls):
        if name in TASK_REGISTRY:
            raise ValueError("Cannot register duplicate task ({})".format(name))
        if not issubclass(cls, FairseqTask):
            raise ValueError("Task ({}: {}) must extend FairseqTask".format(name, cls.__name__))
        if cls.__name__ in TASK_CLASS_NAMES:
            raise ValueError("Cannot register task with duplicate class name ({})".format(cls.__name__))
        TASK_REGISTRY[name] = cls
        TASK_CLASS_NAMES.add(cls.__name__)
        return cls

    return register_task_cls
