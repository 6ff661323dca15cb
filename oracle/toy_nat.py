"""TEST INFRASTRUCTURE ONLY.  A tiny deterministic CMLM-style non-autoregressive model with the interface the reference's
IterativeRefinementGenerator drives (forward_encoder / initialize_output_tokens / forward_decoder / regenerate_length_beam /
encoder.reorder_encoder_out; fairseq/models/nat/cmlm_transformer.py:88-134, nonautoregressive_transformer.py).  The "decoder" is a
fixed random feature map, so the refinement loop, the mask-predict update and the length beam have something real to iterate
on.  `update(logits, tokens, scores, step, max_step) -> (predicted_tokens, tokens, scores)` is pluggable: the golden generator
passes the reference's own `_skeptical_unmasking`, the CPU tests the restatement below, the GPU tests the HIP kernel."""
from collections import namedtuple

import torch

DecoderOut = namedtuple("IterativeRefinementDecoderOut", ["output_tokens", "output_scores", "attn", "step", "max_step", "history"])


class ToyDict:
    def __init__(self, n=40):
        self.n = n

    def bos(self): return 0
    def pad(self): return 1
    def eos(self): return 2
    def unk(self): return 3
    def __len__(self): return self.n


def skeptical_unmasking(scores, nonpad, p):
    """cmlm_transformer.py:19-25 restated: the (n_nonpad - 2) * p lowest scores of every row (stable order)."""
    order = torch.sort(scores, dim=-1, stable=True)[1]
    boundary = ((nonpad.sum(1, keepdim=True).type_as(scores) - 2) * p).long()
    first = torch.arange(scores.size(1), device=scores.device).unsqueeze(0) < boundary
    return torch.zeros_like(first).scatter(1, order, first)


def torch_update(unmask=skeptical_unmasking, unk=3, pad=1):
    def update(logits, tokens, scores, step, max_step):
        masks = tokens.eq(unk)
        sc, tk = torch.log_softmax(logits, dim=-1).max(-1)
        tokens = torch.where(masks, tk.to(tokens.dtype), tokens)
        scores = torch.where(masks, sc, scores)
        predicted = tokens.clone()
        if (step + 1) < max_step:
            sk = unmask(scores, tokens.ne(pad), 1 - (step + 1) / max_step)
            tokens = tokens.masked_fill(sk, unk)
            scores = scores.masked_fill(sk, 0.0)
        return predicted, tokens, scores
    return update


class _Encoder:
    def reorder_encoder_out(self, enc, order):
        return {k: v.index_select(0, order.reshape(-1)) for k, v in enc.items()}


class ToyCMLM:
    allow_length_beam = True

    def __init__(self, d: ToyDict, update, dim=16, seed=0, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.d, self.update, self.device = d, update, device
        self.emb = torch.randn(len(d), dim, generator=g).to(device)
        self.pos = (torch.randn(256, dim, generator=g) * 0.5).to(device)
        self.out = (torch.randn(dim, len(d), generator=g) * 1.5).to(device)
        self.out[:, :4] -= 4.0  # the specials are never predicted
        self.encoder = _Encoder()

    def eval(self):
        return self

    def forward_encoder(self, inputs):
        src, lens = inputs
        keep = src.ne(self.d.pad()).unsqueeze(-1)
        return {"e": (self.emb[src] * keep).sum(1) / lens.unsqueeze(1).float(), "len": lens.clone()}

    def _blank(self, lengths):
        B, T = lengths.size(0), int(lengths.max())
        idx = torch.arange(T, device=lengths.device).unsqueeze(0)
        tok = torch.full((B, T), self.d.pad(), dtype=torch.long, device=lengths.device)
        tok = tok.masked_fill(idx < lengths.unsqueeze(1), self.d.unk())
        tok[:, 0] = self.d.bos()
        tok = tok.scatter(1, (lengths - 1).unsqueeze(1), self.d.eos())
        return DecoderOut(tok, torch.zeros(B, T, device=lengths.device), None, 0, 0, None)

    def initialize_output_tokens(self, enc, src_tokens, target_length=None):
        lengths = target_length if target_length is not None else (enc["len"] + 3).clamp(min=2)
        return self._blank(lengths)

    def regenerate_length_beam(self, decoder_out, beam_size):
        lengths = decoder_out.output_tokens.ne(self.d.pad()).sum(1)
        delta = torch.arange(beam_size, device=lengths.device) - beam_size // 2
        return self._blank((lengths.unsqueeze(1) + delta.unsqueeze(0)).clamp(min=2).reshape(-1))

    def logits(self, tokens, enc, step):
        h = self.emb[tokens]
        h = h + 0.5 * torch.roll(h, 1, dims=1) + 0.5 * torch.roll(h, -1, dims=1) + self.pos[: tokens.size(1)].unsqueeze(0) + enc["e"].unsqueeze(1)
        return (h @ self.out) * (1.0 + 0.3 * step)

    def forward_decoder(self, decoder_out, encoder_out, **kwargs):
        step, max_step = decoder_out.step, decoder_out.max_step
        predicted, tokens, scores = self.update(self.logits(decoder_out.output_tokens, encoder_out, step).float().contiguous(),
                                                decoder_out.output_tokens, decoder_out.output_scores, step, max_step)
        history = decoder_out.history
        if history is not None:
            history.append(predicted.clone())
            if (step + 1) < max_step:
                history.append(tokens.clone())
        return decoder_out._replace(output_tokens=tokens, output_scores=scores, attn=None, history=history)


def toy_sample(d: ToyDict, device="cpu"):
    g = torch.Generator().manual_seed(5)
    lens = torch.tensor([7, 3, 11, 5, 9])
    src = torch.full((5, 11), d.pad(), dtype=torch.long)
    for i, n in enumerate(lens):
        src[i, :n] = torch.randint(4, len(d), (int(n),), generator=g)
    tgt = torch.full((5, 14), d.pad(), dtype=torch.long)
    for i, n in enumerate(lens + 2):
        tgt[i, :n] = torch.randint(4, len(d), (int(n),), generator=g)
    return {"id": torch.arange(5), "net_input": {"src_tokens": src.to(device), "src_lengths": lens.to(device)}, "target": tgt.to(device),
            "ntokens": int((lens + 2).sum())}


SETTINGS = [dict(max_iter=4, adaptive=True), dict(max_iter=3, adaptive=False), dict(max_iter=4, adaptive=True, retain_history=True),
            dict(max_iter=2, adaptive=True, beam_size=3), dict(max_iter=3, adaptive=True, use_true_length=True)]
