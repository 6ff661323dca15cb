"""Mask-predict iterative refinement, the decoding loop downstream of the normalised units: the host control flow
of the reference's `IterativeRefinementGenerator` (fairseq/iterative_refinement_generator.py:18-330) and the per-iteration CMLM
update of its non-autoregressive model (fairseq/models/nat/cmlm_transformer.py:19-134) as one HIP kernel (`dn_cmlm_step`).

Scope: the generator (same constructor arguments, `generate(models, sample)` contract, hypothesis dicts; `speech_source=True` =
the research/TranSpeech variant the S2UT task builds) and the mask-predict update.  The model inside the loop is
`diffnorm_amd.nar_decoder.NARS2UTDecoderModel` (the decoder side of the reference's NARS2UTTransformerModel on the HIP engine; the
speech encoder's output is a given tensor); any other model exposing `forward_encoder / initialize_output_tokens / forward_decoder /
encoder.reorder_encoder_out` (+ `regenerate_length_beam`, `allow_length_beam` for a length beam) plugs in; `cmlm_update` is what
such a model's `forward_decoder` calls after its decoder has produced logits.
"""
from collections import namedtuple
from typing import List, Optional

import torch

DecoderOut = namedtuple("IterativeRefinementDecoderOut", ["output_tokens", "output_scores", "attn", "step", "max_step", "history"])


def _strip_pad(t: torch.Tensor, pad: int) -> torch.Tensor:
    return t[t.ne(pad)]


def _pad_to(t: torch.Tensor, width: int, fill) -> torch.Tensor:
    if t.size(1) >= width:
        return t
    extra = t.new_full((t.size(0), width - t.size(1), *t.shape[2:]), fill)
    return torch.cat([t, extra], dim=1)


class IterativeRefinementGenerator:
    """Same arguments and behaviour as upstream (:19-61): up to `max_iter` + 1 decoder passes; with `adaptive` a sentence is
    finalised as soon as an iteration returns the tokens it was given; finished sentences leave the batch (the encoder output is
    re-ordered to the survivors); with `beam_size` > 1 the length beam's best-scoring candidate is kept (optionally re-ranked)."""

    def __init__(self, tgt_dict, models=None, eos_penalty=0.0, max_iter=10, max_ratio=2, beam_size=1, decoding_format=None,
                 retain_dropout=False, adaptive=True, retain_history=False, reranking=False, use_true_length=False, speech_source=False):
        """speech_source=True: the variant DiffNorm's S2UT task builds (research/TranSpeech/iterative_refinement_generator.py:131-160, task
        speech_to_speech_fasttranslate, fairseq/tasks/nat_s2s_task.py:170-190): the source is a [B, T, 80] feature tensor and the
        model's `initialize_output_tokens(encoder_out, src_lengths)` takes the source LENGTHS (no true-length option)."""
        self.speech_source = speech_source
        self.bos, self.pad, self.unk, self.eos = tgt_dict.bos(), tgt_dict.pad(), tgt_dict.unk(), tgt_dict.eos()
        self.vocab_size = len(tgt_dict)
        self.eos_penalty, self.max_iter, self.max_ratio, self.beam_size = eos_penalty, max_iter, max_ratio, beam_size
        self.reranking, self.decoding_format, self.retain_dropout = reranking, decoding_format, retain_dropout
        self.retain_history, self.adaptive, self.models, self.use_true_length = retain_history, adaptive, models, use_true_length

    # ---- iteration over a dataset (:63-102)
    def generate_batched_itr(self, data_itr, maxlen_a=None, maxlen_b=None, cuda=False, timer=None, prefix_size=0):
        for sample in data_itr:
            if "net_input" not in sample:
                continue
            if timer is not None:
                timer.start()
            with torch.no_grad():
                prefix = sample["target"][:, :prefix_size] if prefix_size > 0 else None
                hypos = self.generate(self.models, sample, prefix_tokens=prefix)
            if timer is not None:
                timer.stop(sample["ntokens"])
            for i, sid in enumerate(sample["id"]):
                yield sid, _strip_pad(sample["net_input"]["src_tokens"][i], self.pad), _strip_pad(sample["target"][i], self.pad), hypos[i]

    # ---- one hypothesis dict (:176-198)
    def _hypothesis(self, step, tokens, scores, attn):
        keep = tokens.ne(self.pad)
        out = {"steps": step, "tokens": tokens[keep], "positional_scores": None, "score": None, "hypo_attn": None, "alignment": None}
        if scores is not None:
            out["positional_scores"] = scores[keep]
            out["score"] = out["positional_scores"].mean()
        if attn is not None:
            out["hypo_attn"] = attn[keep]
            out["alignment"] = out["hypo_attn"].max(dim=1)[1]
        return out

    def _unchanged(self, before, after_out):
        """(:165-174) rows whose tokens did not change; the shorter side is padded so both can be compared (and returned)."""
        tokens, scores, attn = after_out.output_tokens, after_out.output_scores, after_out.attn
        width = max(before.size(1), tokens.size(1))
        before = _pad_to(before, width, self.pad)
        tokens, scores = _pad_to(tokens, width, self.pad), _pad_to(scores, width, 0)
        if attn is not None:
            attn = _pad_to(attn, width, 0)
        return (before == tokens).all(dim=1), after_out._replace(output_tokens=tokens, output_scores=scores, attn=attn)

    @torch.no_grad()
    def generate(self, models, sample, prefix_tokens=None, constraints=None):
        if constraints is not None:
            raise NotImplementedError("Constrained decoding with the IterativeRefinementGenerator is not supported")
        if not self.retain_dropout:
            for m in models:
                m.eval()
        model, reranker = models[0], None
        if self.reranking:
            assert len(models) > 1, "Assuming the last checkpoint is the reranker"
            assert self.beam_size > 1, "Reranking requires multiple translation for each example"
            reranker, models = models[-1], models[:-1]
        if len(models) > 1 and hasattr(model, "enable_ensemble"):
            assert model.allow_ensemble, "{} does not support ensembling".format(model.__class__.__name__)
            model.enable_ensemble(models)

        src_tokens, src_lengths = sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]
        bsz = src_tokens.size(0)
        target_length = None
        if self.use_true_length:
            assert self.beam_size == 1, "beam search is not supported with true length"
            target_length = sample["target"].ne(self.pad).sum(dim=1)

        encoder_out = model.forward_encoder([src_tokens, src_lengths])
        if self.speech_source:
            assert not self.use_true_length, "the speech-source generator has no true-length option"
            state = model.initialize_output_tokens(encoder_out, src_lengths)
        else:
            state = model.initialize_output_tokens(encoder_out, src_tokens, target_length)
        if self.beam_size > 1:
            assert model.allow_length_beam, "{} does not support decoding with length beam.".format(model.__class__.__name__)
            order = torch.arange(bsz, device=src_tokens.device).repeat_interleave(self.beam_size)
            encoder_out = model.encoder.reorder_encoder_out(encoder_out, order)
            state = model.regenerate_length_beam(state, self.beam_size)
            bsz *= self.beam_size

        alive = torch.arange(bsz, device=src_tokens.device)  # original index of every row still being refined
        given = state.output_tokens.clone()
        if self.retain_history:
            state = state._replace(history=[given])
        finalized: List[Optional[list]] = [[] for _ in range(bsz)]
        options = {"eos_penalty": self.eos_penalty, "max_ratio": self.max_ratio, "decoding_format": self.decoding_format}

        for step in range(self.max_iter + 1):
            state = state._replace(step=step, max_step=self.max_iter + 1)
            out = model.forward_decoder(state, encoder_out, **options)
            if self.adaptive:
                done, out = self._unchanged(given, out)
            else:
                done = torch.zeros(out.output_tokens.size(0), dtype=torch.bool, device=out.output_tokens.device)
            if step == self.max_iter:
                done = torch.ones_like(done)
            has_attn = out.attn is not None and out.attn.size(0) > 0
            for row in done.nonzero(as_tuple=False).flatten().tolist():
                hyp = self._hypothesis(step, out.output_tokens[row], out.output_scores[row], out.attn[row] if has_attn else None)
                if self.retain_history:
                    hyp["history"] = [self._hypothesis(step, h[row], None, None) for h in out.history]
                finalized[int(alive[row])] = [hyp]
            if bool(done.all()):
                break
            keep = ~done
            state = out._replace(output_tokens=out.output_tokens[keep], output_scores=out.output_scores[keep],
                                 attn=out.attn[keep] if has_attn else None,
                                 history=[h[keep] for h in out.history] if out.history is not None else None)
            encoder_out = model.encoder.reorder_encoder_out(encoder_out, keep.nonzero(as_tuple=False).squeeze())
            alive = alive[keep]
            given = state.output_tokens.clone()

        if self.beam_size > 1:
            if reranker is not None:
                finalized = self.rerank(reranker, finalized, [src_tokens, src_lengths], self.beam_size)
            best = []
            for i in range(len(finalized) // self.beam_size):  # the length beam's best mean log-probability (:299-312)
                cands = finalized[i * self.beam_size: (i + 1) * self.beam_size]
                scores = torch.stack([c[0]["score"].float().cpu() for c in cands])
                best.append(cands[int(scores.argmax())])
            finalized = best
        return finalized

    def rerank(self, reranker, finalized, encoder_input, beam_size):
        """Autoregressive re-ranking of the length beam (:314-356): candidates are scored by `reranker` teacher-forced on them."""
        def to_batch(cands):
            width = max(c[0]["tokens"].size(0) for c in cands)
            out = cands[0][0]["tokens"].new_full((len(cands), width), self.pad)
            for i, c in enumerate(cands):
                out[i, : c[0]["tokens"].size(0)] = c[0]["tokens"]
            return out

        tokens = to_batch(finalized)
        tokens[:, 0] = self.eos  # teacher forcing starts from eos, as fairseq's translation models do
        enc = reranker.encoder(*encoder_input)
        order = torch.arange(encoder_input[0].size(0), device=tokens.device).repeat_interleave(beam_size)
        enc = reranker.encoder.reorder_encoder_out(enc, order)
        lprobs = reranker.get_normalized_probs(reranker.decoder(tokens[:, :-1], enc), True, None)
        picked = lprobs.gather(2, tokens[:, 1:, None])
        mask = tokens[:, 1:].ne(self.pad)
        picked = picked[:, :, 0].masked_fill_(~mask, 0)
        scores = picked.sum(1) / mask.sum(1).type_as(picked)
        for i in range(len(finalized)):
            finalized[i][0]["score"] = scores[i]
        return finalized


def cmlm_update(logits: torch.Tensor, tokens: torch.Tensor, scores: torch.Tensor, step: int, max_step: int, unk: int, pad: int):
    """The per-iteration update of the CMLM decoder (cmlm_transformer.py:97-127) on the GPU, one kernel: positions that hold
    `unk` (the mask symbol) take argmax / max of log_softmax(logits); then, unless this is the last iteration, the
    (n_nonpad - 2) * (1 - (step + 1) / max_step) lowest-scoring positions are re-masked (`_skeptical_unmasking`, :19-25;
    equal scores are ordered by position).  logits fp32 [B, T, V]; tokens int32 [B, T] and scores fp32 [B, T] are updated in
    place.  Returns (tokens_after_prediction, tokens_after_remasking) like the two `history` entries upstream appends."""
    import ctypes as C  # noqa: F401

    from . import _lib

    lib = _lib.load()
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous()
    assert tokens.dtype == torch.int32 and scores.dtype == torch.float32 and tokens.is_contiguous() and scores.is_contiguous()
    B, T, V = logits.shape
    predicted = torch.empty_like(tokens)
    _lib.check(lib.dn_cmlm_step(logits.data_ptr(), tokens.data_ptr(), scores.data_ptr(), predicted.data_ptr(), B, T, V, int(step),
                                int(max_step), int(unk), int(pad), _lib.current_stream()), "dn_cmlm_step")
    return predicted, tokens
