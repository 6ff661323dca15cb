import torch.nn as nn

from .. import registry


class FairseqCriterion(nn.Module):
    def __init__(self, task):
        super().__init__()
        self.task = task
        tgt = getattr(task, "target_dictionary", None)
        self.padding_idx = tgt.pad() if tgt is not None else -100


build_criterion_, register_criterion, CRITERION_REGISTRY, CRITERION_DATACLASS_REGISTRY = registry.setup_registry(
    "--criterion", base_class=FairseqCriterion, default="cross_entropy")
