"""Speech-unit normalisation driver: the host-side loop of the reference's
research/TranSpeech/diff_norm_synthesis.py (:25-46 unit de-duplication, :132-171 batch assembly, :200-222 sampling
and TSV lines), sharded over ranks by `diffnorm_amd.sharding`."""
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import sharding

TSV_HEADER = "id\tsrc_audio\tsrc_n_frames\ttgt_audio\ttgt_n_frames"


def reduce_token(tokens: Sequence[int]) -> Tuple[List[int], List[int], torch.Tensor]:
    """Run-length de-duplication: (units without consecutive repeats, run lengths, index of each run's first frame)."""
    dedup, durations, keep = [], [], []
    for i, tok in enumerate(tokens):
        if i == 0 or tok != tokens[i - 1]:
            dedup.append(tok)
            keep.append(i)
            durations.append(1)
        else:
            durations[-1] += 1
    return dedup, durations, torch.tensor(keep, dtype=torch.long)


@dataclass
class Utterance:
    audio_id: str
    src_audio: str
    src_n_frames: int
    feat: object                # [T_full, 768] mHuBERT features of the target speech: a tensor, or the path of its .npy file
    tgt_unit: Sequence[int]     # frame-level units (length T_full)
    reduce_tgt_unit: Sequence[int]  # de-duplicated units

    def features(self) -> torch.Tensor:
        """The feature matrix; a path is read when the utterance's batch is assembled (the reference loads per batch too,
        diff_norm_synthesis.py:132-171), so a rank only ever holds the features of the batch it is working on."""
        if isinstance(self.feat, str):
            import numpy as np

            return torch.from_numpy(np.load(self.feat)).float()
        return self.feat


def assemble_batch(items: Sequence[Utterance], device):
    """Selects the first frame of every unit run, zero-pads to the longest utterance (reference :132-171)."""
    feats, units = [], []
    for it in items:
        _, _, keep = reduce_token(list(it.tgt_unit))
        f = it.features()[keep]
        assert f.shape[0] == len(it.reduce_tgt_unit), "reduced units do not match the de-duplicated frames"
        feats.append(f)
        units.append(torch.tensor(list(it.reduce_tgt_unit), dtype=torch.long))
    lens = torch.tensor([f.shape[0] for f in feats], dtype=torch.long)
    B, T = len(items), int(lens.max())
    feat = torch.zeros(B, T, feats[0].shape[1])
    unit = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        feat[b, : lens[b]], unit[b, : lens[b]] = feats[b], units[b]
    # one pinned staging buffer + one async copy per tensor instead of one H2D per utterance (reference :145)
    if torch.cuda.is_available():
        feat, unit = feat.pin_memory(), unit.pin_memory()
    return feat.to(device, non_blocking=True), unit.to(device, non_blocking=True), lens.to(device)


def tsv_line(it: Utterance, pred_units: Sequence[int]) -> str:
    dedup, _, _ = reduce_token(list(pred_units))
    return f"{it.audio_id}\t{it.src_audio}\t{it.src_n_frames}\t{' '.join(str(u) for u in dedup)}\t{len(pred_units)}"


def normalize(ddim_sample: Callable, utterances: Sequence[Utterance], start_step: int = 50, batch_size: int = 100,
              device="cuda:0", group=None) -> Optional[List[str]]:
    """Runs `ddim_sample(feat, input_mask=..., ref_units=..., start_step=...)` (LatentDiscreteModel.ddim_sample) over this
    rank's batches and gathers the TSV lines of all ranks in utterance order (every rank returns the full list)."""
    rank, world = sharding.rank_world(group)
    # A run must give the same results on any number of ranks (and whatever sizes the last batches have).  The 2-byte / f32
    # contractions' fast K order (taps of a causal conv innermost, one staged copy of the rows, on the two 256-row tiles) sums in a
    # different order than the small-batch tiles, so by default a large batch and a small one can differ in the last bit of an fp32
    # sum.  normalize() therefore ALWAYS routes the tap contractions by SHAPE, not by batch size (option "taps_inner" = 2: the
    # 256-row tiles and their order at every batch size) -- the same routing on 1 rank and on N, full speed on the 100-utterance
    # batches this driver forms, 256-row tiles on a short last batch -- for the duration of the call, through the library's
    # option entry (dn_set_option; restored on the way out, no environment mutation), unless the caller has chosen a K order
    # (option set, or DN_TAPS_INNER in the environment at start-up; 0 = term-outer everywhere is the other invariant setting,
    # about 6 % per step slower).  The bf16x3 mode is invariant by construction.
    from . import _lib

    chosen = _lib.get_option("taps_inner")
    batches = sharding.batch_indices(len(utterances), batch_size)
    mine = sharding.my_batches(len(batches), rank, world)
    local = []
    with _lib.option("taps_inner", 2 if chosen is None else chosen):
        for b in mine:
            items = [utterances[i] for i in batches[b]]
            feat, ref_units, lens = assemble_batch(items, device)
            mask = torch.arange(feat.shape[1], device=lens.device).view(1, -1) < lens.view(-1, 1)
            pred, _, _, _ = ddim_sample(feat, input_mask=mask, cond_scale=1.0, ref_units=ref_units, start_step=start_step)
            local.append([tsv_line(it, p.tolist()) for it, p in zip(items, pred)])
    per_batch = sharding.gather_in_order(local, mine, len(batches), group)
    return [line for lines in per_batch for line in lines]
