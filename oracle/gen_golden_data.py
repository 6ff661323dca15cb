"""TEST INFRASTRUCTURE ONLY.  Golden vectors for the on-disk formats either side of the path, produced by the
REAL reference dataset code in the build container:

  reference fairseq/data/audio/repr_to_repr_unit_dataset.py  (_load_samples_from_tsv :309-369, __getitem__ :117-150,
  ordered_indices :178-186, collater :196-258) and fairseq/data/dictionary.py (unit dictionary of speech_decoder_task.py:139-142)

run over a small synthetic corpus (seeded), written to tests/golden/data_formats.npz: the corpus itself (feature arrays, manifest
and TSV lines) plus everything the reference produced from it.  Usage:  python oracle/gen_golden_data.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data_formats.npz")
DIM = 768


def synth_corpus(seed=0):
    """ids, per-utterance (src_feat, tgt_feat, frame-level units); some rows exercise the skip rules."""
    rng = np.random.RandomState(seed)
    ids = ["common_voice_es_%08d" % (19979900 + i) for i in range(7)]
    utts = {}
    for k, uid in enumerate(ids):
        ts, tt = int(rng.randint(3, 9)), int(rng.randint(6, 15))
        units, u = [], int(rng.randint(0, 1000))
        for _ in range(tt):  # runs of repeated units, as k-means units of speech frames have
            if rng.rand() < 0.45:
                u = int(rng.randint(0, 1000))
            units.append(u)
        utts[uid] = (rng.randn(ts, DIM).astype(np.float32), rng.randn(tt, DIM).astype(np.float32), units)
    return ids, utts


def write_corpus(root, split, ids, utts):
    """Lays the corpus out as the reference's tools do: {feat_dir}/{split}.manifest.tsv (first line = directory) +
    {dir}/{id}.feat.npy, and {raw}/{split}.tsv with a header line.  Returns the text of the three files."""
    texts = {}
    for side, pick in (("src_feat", 0), ("tgt_feat", 1)):
        d = os.path.join(root, side, split)
        os.makedirs(d, exist_ok=True)
        lines = [d]
        for uid in ids:
            if side == "tgt_feat" and uid == ids[5]:
                continue  # id missing from one manifest -> the reference skips the row
            np.save(os.path.join(d, uid + ".feat.npy"), utts[uid][pick])
            lines.append("%s.feat.npy\t%d" % (uid, utts[uid][pick].shape[0]))
        lines.insert(3, "")  # blank lines are ignored
        texts[side] = "\n".join(lines) + "\n"
        open(os.path.join(root, side, split + ".manifest.tsv"), "w").write(texts[side])
    raw = os.path.join(root, "raw")
    os.makedirs(raw, exist_ok=True)
    rows = ["id\tsrc_audio\tsrc_n_frames\ttgt_audio\ttgt_n_frames"]
    for uid in ids:
        units = list(utts[uid][2])
        if uid == ids[3]:
            units = units[:-1]  # unit count != feature length -> the reference warns and skips
        rows.append("%s\t%s.mp3\t%d\t%s\t%d" % (uid, uid, 16000 + len(units), " ".join(map(str, units)), len(units)))
    rows.insert(2, "")
    texts["raw"] = "\n".join(rows) + "\n"
    open(os.path.join(raw, split + ".tsv"), "w").write(texts["raw"])
    return texts


class _Cfg:  # the two S2SDataConfig members the reference dataset touches
    shuffle = False

    def get_feature_transforms(self, split, is_train):
        return None

    def get_waveform_transforms(self, split, is_train):
        return None


def main():
    ds_mod, Dictionary = ref_loader.load_reference_dataset()
    ids, utts = synth_corpus()
    out = {}
    with tempfile.TemporaryDirectory() as root:
        texts = write_corpus(root, "dev", ids, utts)
        C = ds_mod.ReprToReprUnitDatasetCreator
        samples = C._load_samples_from_tsv(os.path.join(root, "src_feat"), os.path.join(root, "tgt_feat"),
                                           os.path.join(root, "raw"), "dev")
        d = Dictionary()
        for i in range(1000):
            d.add_symbol(str(i))
        ds = C._from_list("dev", False, samples, _Cfg(), d)
        rel = lambda p: os.path.relpath(p, root)
        out["samples_json"] = json.dumps([{k: (rel(v) if k in (C.KEY_SRC_AUDIO, C.KEY_TGT_AUDIO) else v) for k, v in s.items()}
                                          for s in samples])
        out["ordered_indices"] = np.asarray(ds.ordered_indices())
        out["sizes"] = np.asarray(ds.sizes)
        items = [ds[i] for i in range(len(ds))]
        for i, it in enumerate(items):
            out[f"item{i}_tgt_unit"] = it.tgt_unit.numpy()
            out[f"item{i}_reduce_tgt_unit"] = it.reduce_tgt_unit.numpy()
            out[f"item{i}_reduce_tgt_feat_sum"] = it.reduce_tgt_feat.double().sum(1).numpy()
            dd, dur, keep = ds._reduce_tgt(samples[i][C.KEY_TGT_UNIT])
            out[f"item{i}_durations"] = np.asarray(dur)
            out[f"item{i}_keep"] = keep.numpy()
        order = [int(i) for i in ds.ordered_indices()][:4]
        batch = ds.collater([items[i] for i in order])
        out["batch_order"] = np.asarray(order)
        out["batch_id"] = batch["id"].numpy()
        out["batch_src_tokens"] = batch["net_input"]["src_tokens"].numpy()
        out["batch_src_lengths"] = batch["net_input"]["src_lengths"].numpy()
        for k in ("target", "target_unit", "reduce_target", "reduce_target_unit", "target_lengths", "reduce_target_lengths"):
            out["batch_" + k] = batch[k].numpy()
        out["batch_ntokens"] = np.asarray(batch["ntokens"])
        out["batch_nsentences"] = np.asarray(batch["nsentences"])
        out["dict_len"] = np.asarray(len(d))
        out["dict_specials"] = np.asarray([d.bos(), d.pad(), d.eos(), d.unk()])
        out["dict_encode_probe"] = d.encode_line("0 17 999 1000 x", add_if_not_exist=False, append_eos=False).numpy()
    out["ids"] = np.asarray(ids)
    for uid in ids:
        out["src_" + uid], out["tgt_" + uid] = utts[uid][0], utts[uid][1]
        out["units_" + uid] = np.asarray(utts[uid][2])
    for k, v in texts.items():
        out["text_" + k] = np.asarray(v.replace(root, "{ROOT}"))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(samples), "samples kept of", len(ids))


if __name__ == "__main__":
    main()
