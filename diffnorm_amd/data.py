"""On-disk formats either side of the path and the dataset that feeds it.

Mirrors, by name and behaviour, the reference's
  * feature manifests       examples/textless_nlp/gslm/speech2unit/pretrained/utils.py:127-144
                            `{feat_root}/{split}.manifest.tsv`: first line = feature directory, then `name.feat.npy<TAB>n_frames`;
                            one `[T, 768]` float `.npy` per utterance
  * translation manifest    `{raw}/{split}.tsv`: header, then `id, src_audio, src_n_frames, tgt_audio (space-separated units), tgt_n_frames`
  * sample list             fairseq/data/audio/repr_to_repr_unit_dataset.py:309-369 (`_load_samples_from_tsv`, incl. its skip rules)
  * unit dictionary         fairseq/tasks/speech_decoder_task.py:139-142 on fairseq/data/dictionary.py:20-40 (4 specials, unit u -> u + 4)
  * dataset + collater      repr_to_repr_unit_dataset.py:92-258 (`ReprToReprUnitDataset`)
  * normalisation inputs    research/TranSpeech/diff_norm_synthesis.py:70-130 (`prepare_data`)

Host-side only (NumPy / torch CPU tensors); batches are handed to the engines by `normalize.assemble_batch` or the
criterions.  The difference from upstream is mechanical: `collater(pin=True)` returns pinned tensors so the caller can
issue one asynchronous H2D copy per tensor instead of one blocking copy per utterance.
"""
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .normalize import TSV_HEADER, Utterance, reduce_token

KEY_ID, KEY_SRC_AUDIO, KEY_SRC_N_FRAMES = "id", "src_audio", "src_n_frames"
KEY_TGT_AUDIO, KEY_TGT_N_FRAMES, KEY_TGT_UNIT = "tgt_audio", "tgt_n_frames", "tgt_unit"
EVAL_SAMPLE_CAP = 4000  # upstream keeps 4k (+1, see load_samples) utterances of non-train splits


class UnitDictionary:
    """The task's unit dictionary: fairseq's four specials then the symbols "0".."n_units-1", so unit u has index u + 4
    and index 0 (<s>) doubles as the collater's padding value."""

    BOS, PAD, EOS, UNK = 0, 1, 2, 3

    def __init__(self, n_units: int = 1000):
        self.n_units = n_units
        self.nspecial = 4
        # attribute spellings of fairseq's Dictionary (fairseq/data/dictionary.py:29-33)
        self.bos_index, self.pad_index, self.eos_index, self.unk_index = self.BOS, self.PAD, self.EOS, self.UNK

    def __len__(self):
        return self.n_units + self.nspecial

    def bos(self):
        return self.BOS

    def pad(self):
        return self.PAD

    def eos(self):
        return self.EOS

    def unk(self):
        return self.UNK

    def index(self, sym: str) -> int:
        if sym.isdigit() and str(int(sym)) == sym and int(sym) < self.n_units:
            return int(sym) + self.nspecial
        return self.UNK

    def encode_line(self, line: str, add_if_not_exist: bool = False, append_eos: bool = False) -> torch.Tensor:
        assert not add_if_not_exist, "the unit dictionary is closed"
        words = line.strip().split()  # fairseq.tokenizer.tokenize_line: collapse whitespace, split
        ids = [self.index(w) for w in words] + ([self.EOS] if append_eos else [])
        return torch.tensor(ids, dtype=torch.int32)

    def string(self, indices: Iterable[int]) -> str:
        specials = {self.BOS: "<s>", self.PAD: "<pad>", self.EOS: "</s>", self.UNK: "<unk>"}
        return " ".join(specials.get(int(i), str(int(i) - self.nspecial)) for i in indices)


# ------------------------------------------------------------------------------------------------ manifests
def write_feature_manifest(out_features_path: str, items: Iterable[Tuple[str, np.ndarray]]) -> str:
    """Saves `[T, C]` features as `{out_features_path}/{file_id}.feat.npy` and writes
    `{dirname(out_features_path)}/{basename}.manifest.tsv` (upstream utils.py:127-144).  Returns the manifest path."""
    os.makedirs(out_features_path, exist_ok=True)
    split = os.path.basename(out_features_path)
    manifest = os.path.join(os.path.dirname(out_features_path), f"{split}.manifest.tsv")
    with open(manifest, "w") as fh:
        fh.write(f"{out_features_path}\n")
        for file_id, feats in items:
            name = os.path.basename(file_id).split(".")[0]
            np.save(os.path.join(out_features_path, f"{name}.feat.npy"), np.asarray(feats))
            print(f"{name}.feat.npy\t{feats.shape[0]}", file=fh)
    return manifest


def read_feature_manifest(manifest_file: str) -> Dict[str, Tuple[str, str]]:
    """id -> (feature path, length as written); id = file name up to its first dot (upstream :310-322)."""
    id2feat = {}
    with open(manifest_file, "r") as fh:
        feat_dir = fh.readline().strip()
        for line in fh:
            if len(line.strip()) == 0:
                continue
            feat_name, feat_len = line.strip().split("\t")
            id2feat[feat_name.split(".")[0]] = (f"{feat_dir}/{feat_name}", feat_len)
    return id2feat


def load_samples(src_feat_dir: str, tgt_feat_dir: str, raw_audio_root: str, split: str, log=None) -> List[Dict]:
    """The sample list of a split (upstream `_load_samples_from_tsv`): rows whose id is missing from either feature
    manifest, or whose unit count differs from the target feature length, are skipped; non-train splits stop once more
    than 4000 rows have been kept (so they hold up to 4001, as upstream)."""
    log = log or (lambda msg: None)
    src_id2feat = read_feature_manifest(f"{src_feat_dir}/{split}.manifest.tsv")
    tgt_id2feat = read_feature_manifest(f"{tgt_feat_dir}/{split}.manifest.tsv")
    samples, kept = [], 0
    with open(f"{raw_audio_root}/{split}.tsv") as fh:
        fh.readline()
        for line in fh:
            if len(line.strip()) == 0:
                continue
            src_id, _src_audio, _src_n, tgt_audio_token, _tgt_n = line.rstrip().split("\t")
            if src_id not in src_id2feat or src_id not in tgt_id2feat:
                log(f"src_id: {src_id} not found in feat manifest")
                continue
            src_feat_path, src_feat_len = src_id2feat[src_id]
            tgt_feat_path, tgt_feat_len = tgt_id2feat[src_id]
            tgt_tokens = [int(x) for x in tgt_audio_token.split(" ")]
            if len(tgt_tokens) != int(tgt_feat_len):
                log(f"warning: mismatched feature and unit size. tgt_tokens: {len(tgt_tokens)}, tgt_feat_len: {tgt_feat_len}")
                continue
            samples.append({KEY_ID: src_id, KEY_SRC_AUDIO: src_feat_path, KEY_SRC_N_FRAMES: src_feat_len,
                            KEY_TGT_AUDIO: tgt_feat_path, KEY_TGT_UNIT: tgt_tokens, KEY_TGT_N_FRAMES: tgt_feat_len})
            kept += 1
            if "train" not in split and kept > EVAL_SAMPLE_CAP:
                break
    return samples


# ------------------------------------------------------------------------------------------------ dataset
@dataclass
class ReprToReprDatasetItem:
    index: int
    src_feat: torch.Tensor         # [Ts, C]
    tgt_feat: torch.Tensor         # [Tt, C]
    tgt_unit: torch.Tensor         # [Tt] dictionary indices (unit + 4)
    reduce_tgt_unit: torch.Tensor  # [Tr] de-duplicated
    reduce_tgt_feat: torch.Tensor  # [Tr, C] feature of the first frame of every unit run


class ReprToReprUnitDataset(torch.utils.data.Dataset):
    """Source-feature -> (target feature, unit) pairs for VAE / diffusion training and evaluation."""

    def __init__(self, split: str, is_train_split: bool, samples: Sequence[Dict], tgt_dict: Optional[UnitDictionary] = None,
                 shuffle: bool = False):
        self.split, self.is_train_split = split, is_train_split
        self.audio_paths = [s[KEY_SRC_AUDIO] for s in samples]
        self.tgt_feat_paths = [s[KEY_TGT_AUDIO] for s in samples]
        self.tgt_units = [s[KEY_TGT_UNIT] for s in samples]
        self.ids = [s[KEY_ID] for s in samples]
        self.src_n_frames = [int(s[KEY_SRC_N_FRAMES]) for s in samples]
        self.tgt_n_frames = [int(s[KEY_TGT_N_FRAMES]) for s in samples]
        self.n_samples = len(samples)
        self.tgt_dict = tgt_dict or UnitDictionary()
        self.shuffle = shuffle if is_train_split else False

    @classmethod
    def from_tsv(cls, src_feat_dir, tgt_feat_dir, audio_root, split, is_train_split, tgt_dict=None, shuffle=False):
        return cls(split, is_train_split, load_samples(src_feat_dir, tgt_feat_dir, audio_root, split), tgt_dict, shuffle)

    def __len__(self):
        return self.n_samples

    def __getitem__(self, index: int) -> ReprToReprDatasetItem:
        src_feat = torch.from_numpy(np.load(self.audio_paths[index])).float()
        tgt_feat = torch.from_numpy(np.load(self.tgt_feat_paths[index])).float()
        units = self.tgt_units[index]
        reduced, _, keep = reduce_token(units)
        enc = lambda seq: self.tgt_dict.encode_line(" ".join(str(x) for x in seq)).long()
        return ReprToReprDatasetItem(index, src_feat, tgt_feat, enc(units), enc(reduced), tgt_feat[keep])

    def num_tokens(self, index):
        return self.tgt_n_frames[index]

    def size(self, index):
        return self.tgt_n_frames[index]

    @property
    def sizes(self):
        return np.array(self.tgt_n_frames)

    def ordered_indices(self):
        """Longest target first; ties in original (or, when shuffling, random) order."""
        first = np.random.permutation(len(self)) if self.shuffle else np.arange(len(self))
        return np.lexsort([first, [-n for n in self.tgt_n_frames]])

    def collater(self, samples: List[ReprToReprDatasetItem], pin: bool = False) -> Dict:
        """Zero-padded batch, rows ordered by descending source length (upstream :196-258): the criterion's sample dict."""
        if len(samples) == 0:
            return {}
        B, C = len(samples), samples[0].src_feat.shape[1]
        indices = torch.tensor([x.index for x in samples], dtype=torch.long)
        src_lengths = torch.tensor([x.src_feat.shape[0] for x in samples], dtype=torch.long)
        tgt_lengths = torch.tensor([x.tgt_feat.shape[0] for x in samples], dtype=torch.long)
        red_lengths = torch.tensor([x.reduce_tgt_unit.shape[0] for x in samples], dtype=torch.long)
        src_lengths, order = src_lengths.sort(descending=True)
        order_l = order.tolist()

        def padded(get, length, trailing=(), dtype=torch.float32):
            out = torch.zeros((B, int(length)) + tuple(trailing), dtype=dtype, pin_memory=pin and torch.cuda.is_available())
            for row, i in enumerate(order_l):  # written in final row order: no second pass to permute
                v = get(samples[i])
                out[row, : v.shape[0]] = v
            return out

        src = padded(lambda s: s.src_feat, src_lengths.max(), (C,))
        tgt = padded(lambda s: s.tgt_feat, tgt_lengths.max(), (C,))
        tgt_unit = padded(lambda s: s.tgt_unit, tgt_lengths.max(), dtype=torch.long)
        red_unit = padded(lambda s: s.reduce_tgt_unit, red_lengths.max(), dtype=torch.long)
        red_feat = padded(lambda s: s.reduce_tgt_feat, red_lengths.max(), (C,))
        red_lengths = red_lengths.index_select(0, order)
        return {
            "id": indices.index_select(0, order),
            "net_input": {"src_tokens": src, "src_lengths": src_lengths, "prev_output_tokens": None, "tgt_speaker": None},
            "speaker": None,
            "target": tgt,
            "target_unit": tgt_unit,
            "reduce_target": red_feat,
            "reduce_target_unit": red_unit,
            "target_lengths": tgt_lengths.index_select(0, order),
            "reduce_target_lengths": red_lengths,
            "ntokens": red_lengths.sum().item(),
            "nsentences": B,
        }


# ------------------------------------------------------------------------------------------------ normalisation I/O
def read_unit_tsv(path: str) -> Dict[str, Tuple[str, int, str, int]]:
    """`id -> (src_audio, src_n_frames, units string, n_frames)` of a 5-column unit TSV; malformed rows are dropped
    (upstream diff_norm_synthesis.py:79-88)."""
    rows = {}
    with open(path, "r") as fh:
        fh.readline()
        for line in fh:
            cols = line.strip().split("\t")
            if len(cols) != 5:
                continue
            audio_id, src_audio, src_n, tgt_audio, tgt_n = cols
            rows[audio_id] = (src_audio, int(src_n), tgt_audio, int(tgt_n))
    return rows


def load_normalization_inputs(reduce_tsv_dir: str, orig_tsv_dir: str, feature_dir: str, split: str) -> List[Utterance]:
    """The utterance list `diff_norm_synthesis.prepare_data` builds (:70-118): rows of the original (frame-level) TSV
    that also appear in the reduced TSV and have a feature file, in the original TSV's order.  Only the feature PATHS are kept:
    `normalize.assemble_batch` reads the files of one batch at a time (and only on the rank that owns the batch) and stages
    them through one pinned buffer."""
    reduced = read_unit_tsv(f"{reduce_tsv_dir}/{split}.tsv")
    out = []
    with open(f"{orig_tsv_dir}/{split}.tsv", "r") as fh:
        fh.readline()
        for line in fh:
            cols = line.strip().split("\t")
            if len(cols) != 5:
                continue
            audio_id, _, _, tgt_audio, _ = cols
            feature_file = f"{feature_dir}/{split}/{audio_id}.feat.npy"
            if audio_id not in reduced or not os.path.exists(feature_file):
                continue
            src_audio, src_n, red_units, red_n = reduced[audio_id]
            red = [int(u) for u in red_units.split(" ")]
            assert len(red) == red_n, f"{audio_id}: reduced TSV says {red_n} units, row holds {len(red)}"
            out.append(Utterance(audio_id, src_audio, src_n, feature_file, [int(u) for u in tgt_audio.split(" ")], red))
    return out


def write_unit_tsv(path: str, lines: Iterable[str]) -> None:
    """Writes the normalised-unit TSV (upstream :200-222): header + one `normalize.tsv_line` per utterance."""
    with open(path, "w") as fh:
        fh.write(TSV_HEADER + "\n")
        for line in lines:
            print(line, file=fh)
