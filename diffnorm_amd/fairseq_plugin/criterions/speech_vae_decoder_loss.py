"""`--criterion speech_vae_decoder_loss` (reference fairseq/criterions/speech_vae_decoder_loss.py:14-129):
0.1 * LSCE(eps=0.1)/ntokens + 10 * MSE + 1e-4 * KL, sample_size = nsentences."""
from typing import Any, Dict, List

import torch

from ...latent_module import label_smoothed_nll_loss
from ..registry import FairseqCriterion, register_criterion


def _weighted(logging_outputs, keys):
    ns = [log.get("sample_size", 0) for log in logging_outputs]
    ntot = sum(ns)
    ws = [n / (ntot + 1e-8) for n in ns]
    out = {k: sum(log.get(k, 0) * w for log, w in zip(logging_outputs, ws)) for k in keys}
    out["sample_size"] = ntot
    return out


@register_criterion("speech_vae_decoder_loss")
class SpeechVAEDecoderLoss(FairseqCriterion):
    def __init__(self, task):
        super().__init__(task)
        self.eps = 0.1

    def forward(self, model, sample, reduction="mean"):
        model_kwargs = dict(src_feature=sample["net_input"]["src_tokens"], src_lengths=sample["net_input"]["src_lengths"],
                            tgt_lengths=sample["reduce_target_lengths"], unk_token=self.task.tgt_dict.unk_index)
        unit = sample["reduce_target_unit"]
        if sample.get("posterior_noise") is not None:  # parity runs inject the draw the reference makes on the CPU generator
            model_kwargs["noise"] = sample["posterior_noise"]
        if getattr(getattr(model, "encoder", None), "_train_engine", None) is not None:
            # HIP training engine: LS-CE, masked MSE and KL and their gradients are kernels of the engine (the same algebra as
            # below, :60-83); the loss tensor is autograd-connected to the flat parameter through the HIP backward
            stats, _ = model(sample["reduce_target"], unit, return_stats=True, ntokens=sample["ntokens"], **model_kwargs)
            host = stats.tolist()  # one read for the logging output, like upstream's utils.item calls
            sample_size = sample["nsentences"]
            logging_output = {"loss": host[0], "nll_loss": host[1], "mse_loss": host[2], "kl_loss": host[3], "acc": host[4],
                              "ntokens": sample["ntokens"], "nsentences": sample["nsentences"], "sample_size": sample_size}
            return stats[0], sample_size, logging_output
        mse_loss, lm_pred, kl_loss = model(sample["reduce_target"], unit, **model_kwargs)
        lprobs = model.get_normalized_probs([lm_pred], log_probs=True)
        lprobs = lprobs.view(-1, lprobs.size(-1))
        target = unit.to(lprobs.device).view(-1)
        keep = target.ne(0)
        acc = torch.sum(lprobs.argmax(1).masked_select(keep).eq(target.masked_select(keep))) / torch.sum(keep)
        loss, nll_loss = label_smoothed_nll_loss(lprobs, target, self.eps, ignore_index=0, reduce=True)
        loss, nll_loss = loss / sample["ntokens"], nll_loss / sample["ntokens"]
        loss = 0.1 * loss + 10 * mse_loss + 0.0001 * kl_loss
        sample_size = sample["nsentences"]
        logging_output = {"loss": loss.item(), "nll_loss": nll_loss.item(), "mse_loss": mse_loss.item(), "kl_loss": kl_loss.item(),
                          "acc": acc.item(), "ntokens": sample["ntokens"], "nsentences": sample["nsentences"],
                          "sample_size": sample_size}
        return loss, sample_size, logging_output

    @classmethod
    def reduce_metrics(cls, logging_outputs: List[Dict[str, Any]]):
        """Sample-size-weighted means, as upstream (:97-112); returned as a dict (and logged when fairseq is present)."""
        agg = _weighted(logging_outputs, ["loss", "nll_loss", "mse_loss", "kl_loss", "acc"])
        try:  # pragma: no cover
            from fairseq import metrics

            for k in ("loss", "nll_loss", "mse_loss", "kl_loss", "acc"):
                metrics.log_scalar(k, agg[k], agg["sample_size"], round=3)
            metrics.log_scalar("sample_size", agg["sample_size"], len(logging_outputs))
        except ImportError:
            pass
        return agg

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return False
