"""`--task speech_decoder` (reference fairseq/tasks/speech_decoder_task.py:33-259): unit dictionary, model /
criterion construction and the train / valid step contract around the HIP-backed models.

`load_dataset` reads the reference's manifests (`diffnorm_amd.data`: `{feat_dir}/{split}.manifest.tsv` + `{data}/{split}.tsv`)
when `--src-feat-dir/--tgt-feat-dir` are given, else serves the synthetic (feat, unit) pairs the BASELINE configs
use.  `train_step` is the reference's: criterion forward, then `optimizer.backward(loss)` -- for the speech VAE that is the HIP
training engine's backward.
"""
import zlib

import torch

from ...data import ReprToReprUnitDataset, UnitDictionary
from ...latent_module import wrapped_by_torch_ddp
from ...profiling import profile_range
from ..registry import MODEL_REGISTRY, ARCH_MODEL_REGISTRY, ARCH_CONFIG_REGISTRY, CRITERION_REGISTRY, FairseqTask, register_task


class SyntheticReprUnitDataset(torch.utils.data.Dataset):
    """Synthetic stand-in for ReprToReprUnitDataset producing the criterion's sample dict
    (reference fairseq/data/audio/repr_to_repr_unit_dataset.py:239-258): N(0,1) features, units U{4..1003}, 0 = pad."""

    def __init__(self, n, min_len, max_len, dim=768, vocab=1004, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.lens = torch.randint(min_len, max_len + 1, (n,), generator=g)
        self.dim, self.vocab, self.seed = dim, vocab, seed

    def __len__(self):
        return len(self.lens)

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        T = int(self.lens[i])
        return {"id": i, "feat": torch.randn(T, self.dim, generator=g), "unit": torch.randint(4, self.vocab, (T,), generator=g)}

    def collater(self, items):
        B, T = len(items), max(it["feat"].shape[0] for it in items)
        feat = torch.zeros(B, T, self.dim)
        unit = torch.zeros(B, T, dtype=torch.long)
        lens = torch.tensor([it["feat"].shape[0] for it in items])
        for b, it in enumerate(items):
            feat[b, : lens[b]], unit[b, : lens[b]] = it["feat"], it["unit"]
        return {"id": torch.tensor([it["id"] for it in items]),
                "net_input": {"src_tokens": feat, "src_lengths": lens},
                "target": feat, "target_unit": unit, "target_lengths": lens,
                "reduce_target": feat, "reduce_target_unit": unit, "reduce_target_lengths": lens,
                "ntokens": int(lens.sum()), "n_units": int((unit != 0).sum()), "nsentences": B}


def _check_optimizer_holds_flat_params(enc, optimizer):
    """The HIP training engine's ONE parameter must be what the optimizer updates.  An optimizer that was built from
    `model.parameters()` BEFORE the switch to the training engine holds the deleted per-tensor parameters and would train
    nothing, silently: refuse that here.  (Optimizers over the engine's flat buffers -- optim.FlatOptimizer -- carry `engine`.)"""
    flat = getattr(getattr(enc, "model", enc), "flat_params", None)
    if flat is None or optimizer is None:
        return
    eng = getattr(optimizer, "engine", None)
    if eng is not None:
        if eng is not enc._train_engine:
            raise RuntimeError("the optimizer was built over another training engine than the model's")
        return
    params = getattr(optimizer, "params", None)  # FairseqOptimizer.params
    if params is None and hasattr(optimizer, "param_groups"):
        params = [p for g in optimizer.param_groups for p in g["params"]]
    if params is not None and not any(p is flat for p in params):
        raise RuntimeError("the optimizer does not hold the HIP training engine's `flat_params`: it was built before the model switched to "
                           "the training engine.  Build the model from a training namespace (--optimizer set, or hip_train_engine=True) so that "
                           "model.to(device) makes the switch, or call model.encoder.enable_training() before building the optimizer")


class _SpeechTaskBase(FairseqTask):
    @classmethod
    def add_args(cls, parser):
        parser.add_argument("data", help="manifest root path")
        parser.add_argument("--config-yaml", type=str, default="config.yaml")
        parser.add_argument("--max-source-positions", default=6000, type=int)
        parser.add_argument("--max-target-positions", default=1024, type=int)
        parser.add_argument("--target-is-code", action="store_true")
        parser.add_argument("--target-code-size", type=int, default=None, help="# discrete units")
        parser.add_argument("--save-audio", action="store_true")
        parser.add_argument("--n-frames-per-step", type=int, default=1)
        parser.add_argument("--eval-inference", action="store_true")
        parser.add_argument("--eval-args", type=str, default="{}")
        parser.add_argument("--eos-prob-threshold", type=float, default=0.5)
        parser.add_argument("--mcd-normalize-type", type=str, default="targ")
        parser.add_argument("--vocoder", type=str, default="griffin_lim")
        parser.add_argument("--spec-bwd-max-iter", type=int, default=8)
        parser.add_argument("--infer-target-lang", type=str, default="")
        parser.add_argument("--dummy-config", type=str)
        parser.add_argument("--vocoder-config", type=str)
        parser.add_argument("--tgt-feat-dir", type=str)
        parser.add_argument("--src-feat-dir", type=str)

    def __init__(self, args, tgt_dict):
        super().__init__(args)
        self.args = args
        self.tgt_dict = tgt_dict

    @classmethod
    def setup_task(cls, args, **kwargs):
        n_units = getattr(args, "target_code_size", None) or 1000
        return cls(args, UnitDictionary(n_units))

    @property
    def target_dictionary(self):
        return self.tgt_dict

    @property
    def source_dictionary(self):
        return None

    def max_positions(self):
        return getattr(self.args, "max_source_positions", 6000), getattr(self.args, "max_target_positions", 1024)

    def load_dataset(self, split, epoch=1, combine=False, **kwargs):
        src_dir, tgt_dir = getattr(self.args, "src_feat_dir", None), getattr(self.args, "tgt_feat_dir", None)
        if src_dir and tgt_dir:  # reference speech_decoder_task.py:161-173
            self.datasets[split] = ReprToReprUnitDataset.from_tsv(src_dir, tgt_dir, self.args.data, split,
                                                                  is_train_split=split.startswith("train"),
                                                                  tgt_dict=self.tgt_dict, shuffle=True)
            return self.datasets[split]
        n = kwargs.get("n", 64)
        self.datasets[split] = SyntheticReprUnitDataset(n, kwargs.get("min_len", 64), kwargs.get("max_len", 512),
                                                        vocab=len(self.tgt_dict), seed=zlib.crc32(split.encode()) % 1000)  # stable across processes / ranks
        return self.datasets[split]

    def build_model(self, args, from_checkpoint=False):
        arch = args.arch
        ARCH_CONFIG_REGISTRY[arch](args)
        return ARCH_MODEL_REGISTRY[arch].build_model(args, self)

    def build_criterion(self, args):
        return CRITERION_REGISTRY[args.criterion](self)

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
        """reference speech_decoder_task.py:212-225: forward through the criterion, `optimizer.backward(loss)`.  The model's
        VAE is switched to the HIP training engine on first use (its flat master buffer becomes the parameter the optimizer
        sees); the backward pass behind `loss.backward()` is dn_vae_train_backward."""
        model.train()
        enc = getattr(model, "encoder", None)
        if hasattr(enc, "enable_training"):
            enc.enable_training()  # (a training run's model did this when it was moved to the GPU: models/common_args.is_training_run)
            _check_optimizer_holds_flat_params(enc, optimizer)
            # fairseq's default `--ddp-backend pytorch_ddp` hands the task the model inside torch's DistributedDataParallel, whose
            # reducer only sees gradients that come out of the autograd graph: the bridge then returns them instead of writing the
            # engine's buffer in place (latent_module._finish_backward).  legacy_ddp / one rank: the in-place form.
            through = wrapped_by_torch_ddp(model)
            if through and not getattr(enc, "_grads_through_autograd", False):
                enc._train_engine.work_current = False  # DDP's constructor broadcast rank 0's parameters into the master buffer
            enc._grads_through_autograd = through
        with profile_range("forward"):  # the reference's range names (speech_decoder_task.py:215-220)
            loss, sample_size, logging_output = criterion(model, sample)
        if ignore_grad:
            loss = loss * 0
        with profile_range("backward"):
            optimizer.backward(loss)
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            return criterion(model, sample)

    def valid_step_with_inference(self, sample, model, generator):
        pass  # as upstream (speech_decoder_task.py:241-242)

    def optimizer_step(self, optimizer, model, update_num):
        optimizer.step()


@register_task("speech_decoder")
class SpeechDecoderTask(_SpeechTaskBase):
    pass
