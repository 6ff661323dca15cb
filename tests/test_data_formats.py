"""manifests, unit TSVs, the unit dictionary, ReprToReprUnitDataset and its collater against golden vectors the
REAL reference dataset code produced (oracle/gen_golden_data.py -> tests/golden/data_formats.npz).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from diffnorm_amd import data as D
from diffnorm_amd import normalize as N


@pytest.fixture(scope="module")
def corpus(golden, tmp_path_factory):
    g = golden("data_formats")
    root = str(tmp_path_factory.mktemp("corpus"))
    ids = [str(x) for x in g["ids"]]
    for side in ("src_feat", "tgt_feat"):
        os.makedirs(os.path.join(root, side, "dev"))
        open(os.path.join(root, side, "dev.manifest.tsv"), "w").write(str(g["text_" + side]).replace("{ROOT}", root))
    os.makedirs(os.path.join(root, "raw"))
    open(os.path.join(root, "raw", "dev.tsv"), "w").write(str(g["text_raw"]))
    for uid in ids:
        np.save(os.path.join(root, "src_feat", "dev", uid + ".feat.npy"), g["src_" + uid])
        np.save(os.path.join(root, "tgt_feat", "dev", uid + ".feat.npy"), g["tgt_" + uid])
    return g, root, ids


def test_unit_dictionary_matches_fairseq_dictionary(golden):
    g = golden("data_formats")
    d = D.UnitDictionary(1000)
    assert len(d) == int(g["dict_len"])
    assert [d.bos(), d.pad(), d.eos(), d.unk()] == g["dict_specials"].tolist()
    assert d.encode_line("0 17 999 1000 x").tolist() == g["dict_encode_probe"].tolist()
    assert d.encode_line("5  6\t7", append_eos=True).tolist() == [9, 10, 11, 2]
    assert d.string([4, 1003, 3, 0]) == "0 999 <unk> <s>"
    assert d.index("007") == d.unk()  # the symbol table holds "7", not "007"


def test_load_samples_skip_rules_and_layout(corpus):
    g, root, ids = corpus
    msgs = []
    samples = D.load_samples(f"{root}/src_feat", f"{root}/tgt_feat", f"{root}/raw", "dev", log=msgs.append)
    ref = json.loads(str(g["samples_json"]))
    assert len(samples) == len(ref) == 5
    for s, r in zip(samples, ref):
        assert s[D.KEY_ID] == r["id"] and s[D.KEY_TGT_UNIT] == r["tgt_unit"]
        assert s[D.KEY_SRC_N_FRAMES] == r["src_n_frames"] and s[D.KEY_TGT_N_FRAMES] == r["tgt_n_frames"]  # kept as strings
        assert os.path.relpath(s[D.KEY_SRC_AUDIO], root) == r["src_audio"]
        assert os.path.relpath(s[D.KEY_TGT_AUDIO], root) == r["tgt_audio"]
    assert any("not found in feat manifest" in m for m in msgs) and any("mismatched feature and unit size" in m for m in msgs)


def test_dataset_items_order_and_collater(corpus):
    g, root, ids = corpus
    ds = D.ReprToReprUnitDataset.from_tsv(f"{root}/src_feat", f"{root}/tgt_feat", f"{root}/raw", "dev", is_train_split=False)
    assert len(ds) == 5 and ds.sizes.tolist() == g["sizes"].tolist()
    assert ds.ordered_indices().tolist() == g["ordered_indices"].tolist()
    items = [ds[i] for i in range(len(ds))]
    for i, it in enumerate(items):
        assert it.tgt_unit.dtype == torch.long and it.tgt_unit.tolist() == g[f"item{i}_tgt_unit"].tolist()
        assert it.reduce_tgt_unit.tolist() == g[f"item{i}_reduce_tgt_unit"].tolist()
        np.testing.assert_array_equal(it.reduce_tgt_feat.double().sum(1).numpy(), g[f"item{i}_reduce_tgt_feat_sum"])
        dedup, dur, keep = N.reduce_token(ds.tgt_units[i])
        assert dur == g[f"item{i}_durations"].tolist() and keep.tolist() == g[f"item{i}_keep"].tolist()
        assert [u + 4 for u in dedup] == it.reduce_tgt_unit.tolist()
    batch = ds.collater([items[i] for i in g["batch_order"].tolist()])
    assert batch["id"].tolist() == g["batch_id"].tolist()
    np.testing.assert_array_equal(batch["net_input"]["src_tokens"].numpy(), g["batch_src_tokens"])
    assert batch["net_input"]["src_lengths"].tolist() == g["batch_src_lengths"].tolist()
    for k in ("target", "target_unit", "reduce_target", "reduce_target_unit", "target_lengths", "reduce_target_lengths"):
        np.testing.assert_array_equal(batch[k].numpy(), g["batch_" + k], err_msg=k)
    assert batch["ntokens"] == int(g["batch_ntokens"]) and batch["nsentences"] == int(g["batch_nsentences"])
    assert batch["net_input"]["prev_output_tokens"] is None and batch["speaker"] is None
    assert ds.collater([]) == {}


def test_feature_manifest_round_trip(tmp_path):
    rng = np.random.RandomState(3)
    feats = [("clip_a.wav", rng.randn(5, 768).astype(np.float32)), ("/some/dir/clip_b.flac", rng.randn(9, 768).astype(np.float32))]
    out_dir = str(tmp_path / "feats" / "test")
    manifest = D.write_feature_manifest(out_dir, feats)
    assert manifest == str(tmp_path / "feats" / "test.manifest.tsv")
    assert open(manifest).read() == f"{out_dir}\nclip_a.feat.npy\t5\nclip_b.feat.npy\t9\n"
    back = D.read_feature_manifest(manifest)
    assert back == {"clip_a": (f"{out_dir}/clip_a.feat.npy", "5"), "clip_b": (f"{out_dir}/clip_b.feat.npy", "9")}
    np.testing.assert_array_equal(np.load(back["clip_b"][0]), feats[1][1])


def test_normalization_inputs_and_unit_tsv(tmp_path):
    """prepare_data's joins (reduced TSV x original TSV x feature files) and the normalised-unit TSV a run writes."""
    rng = np.random.RandomState(5)
    full = {"u1": [3, 3, 9, 9, 9, 2], "u2": [7, 7, 7], "u3": [1, 2], "u4": [5, 5]}
    for d in ("orig", "reduce", "feat/dev"):
        os.makedirs(tmp_path / d)
    with open(tmp_path / "orig" / "dev.tsv", "w") as fo, open(tmp_path / "reduce" / "dev.tsv", "w") as fr:
        fo.write(N.TSV_HEADER + "\n")
        fr.write(N.TSV_HEADER + "\n")
        for uid, units in full.items():
            fo.write(f"{uid}\t{uid}.mp3\t100\t{' '.join(map(str, units))}\t{len(units)}\n")
            red, _, _ = N.reduce_token(units)
            if uid != "u3":  # u3 missing from the reduced TSV
                fr.write(f"{uid}\t{uid}.mp3\t100\t{' '.join(map(str, red))}\t{len(red)}\n")
            if uid != "u4":  # u4 has no feature file
                np.save(tmp_path / "feat" / "dev" / f"{uid}.feat.npy", rng.randn(len(units), 768).astype(np.float32))
        fo.write("broken row without tabs\n")
    utts = D.load_normalization_inputs(str(tmp_path / "reduce"), str(tmp_path / "orig"), str(tmp_path / "feat"), "dev")
    assert [u.audio_id for u in utts] == ["u1", "u2"]
    assert list(utts[0].reduce_tgt_unit) == [3, 9, 2] and utts[0].features().shape == (6, 768) and isinstance(utts[0].feat, str) and utts[0].src_n_frames == 100

    def fake_ddim_sample(feat, input_mask, cond_scale, ref_units, start_step):  # echoes the reference units
        lens = input_mask.sum(1).tolist()
        return [ref_units[b, : lens[b]] for b in range(feat.shape[0])], 0, 0, feat

    lines = N.normalize(fake_ddim_sample, utts, start_step=5, batch_size=2, device="cpu")
    D.write_unit_tsv(str(tmp_path / "out.tsv"), lines)
    rows = D.read_unit_tsv(str(tmp_path / "out.tsv"))
    assert rows == {"u1": ("u1.mp3", 100, "3 9 2", 3), "u2": ("u2.mp3", 100, "7", 1)}


def test_plugin_task_loads_the_manifest_dataset(corpus):
    """`--task speech_decoder` with --src-feat-dir/--tgt-feat-dir reads the reference's manifests (speech_decoder_task.py:161-173)."""
    import argparse

    from diffnorm_amd.fairseq_plugin.tasks.speech_decoder_task import SpeechDecoderTask

    g, root, ids = corpus
    args = argparse.Namespace(data=f"{root}/raw", src_feat_dir=f"{root}/src_feat", tgt_feat_dir=f"{root}/tgt_feat",
                              target_is_code=True, target_code_size=1000)
    task = SpeechDecoderTask.setup_task(args)
    assert len(task.target_dictionary) == 1004
    ds = task.load_dataset("dev")
    assert isinstance(ds, D.ReprToReprUnitDataset) and len(ds) == 5 and ds.shuffle is False
    batch = ds.collater([ds[i] for i in ds.ordered_indices()[:2]])
    assert batch["target"].shape[0] == 2 and batch["target_unit"].min().item() >= 0 and batch["nsentences"] == 2
