"""`--criterion ddpm_discrete_loss` (reference fairseq/criterions/ddpm_discrete_loss.py:14-109): passes the model's
loss dict through, sample_size = nsentences."""
from typing import Any, Dict, List

from ..registry import FairseqCriterion, register_criterion
from .speech_vae_decoder_loss import _weighted


@register_criterion("ddpm_discrete_loss")
class DDPMDiscreteLoss(FairseqCriterion):
    def __init__(self, task):
        super().__init__(task)
        self.eps = 0.2

    def forward(self, model, sample, reduction="mean"):
        model_kwargs = dict(src_feature=sample["net_input"]["src_tokens"], src_lengths=sample["net_input"]["src_lengths"],
                            tgt_lengths=sample["reduce_target_lengths"], unk_token=self.task.tgt_dict.unk_index)
        model_kwargs.update(sample.get("diffusion_draws") or {})  # parity runs inject t and the three noise tensors the reference draws
        d = model(sample["reduce_target"], sample["reduce_target_unit"], **model_kwargs)
        loss = d["total_loss"]
        sample_size = sample["nsentences"]
        logging_output = {"loss": loss.item(), "noise_loss": d["noise_loss"].item(), "nll_loss": d["nll_loss"].item(),
                          "mse_loss": d["recon_mse_loss"].item(), "acc": d["acc"].item(), "ntokens": sample["ntokens"],
                          "nsentences": sample["nsentences"], "sample_size": sample_size}
        return loss, sample_size, logging_output

    @classmethod
    def reduce_metrics(cls, logging_outputs: List[Dict[str, Any]]):
        agg = _weighted(logging_outputs, ["loss", "noise_loss", "mse_loss", "nll_loss", "acc"])
        try:  # pragma: no cover
            from fairseq import metrics

            for k in ("loss", "noise_loss", "mse_loss", "nll_loss", "acc"):
                metrics.log_scalar(k, agg[k], agg["sample_size"], round=3)
            metrics.log_scalar("sample_size", agg["sample_size"], len(logging_outputs))
        except ImportError:
            pass
        return agg

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return False
