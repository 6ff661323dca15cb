"""Checkpoints in the reference layout (fairseq/checkpoint_utils.py:35-186, fairseq/trainer.py:392-436): the trainer's top-level
keys, `state["model"]` under the reference's parameter names (`encoder.*`; diffusion: `encoder.model.*` +
`encoder.speech_decoder.*`), round trip through a fresh model, and the plugin's VAE loader
(`diff_discrete._load_speech_decoder`, reference diff_discrete.py:70-78).  CPU only: the mirror modules own their parameters
without a device."""
import types

import torch

import diffnorm_oracle as O
from gen_golden_configs import CHAIN_EPS, CHAIN_VAE


def _vae(seed=1):
    from diffnorm_amd.latent_module import SpeechVAEEncoderDecoder

    return SpeechVAEEncoderDecoder(dim=CHAIN_VAE.dim, latent_dim=CHAIN_VAE.latent_dim, dtype="f32", seed=seed)


def test_vae_checkpoint_round_trip_and_reference_keys(tmp_path):
    from diffnorm_amd import checkpoint
    from diffnorm_amd.fairseq_plugin.registry import FairseqEncoderModel

    model = FairseqEncoderModel(_vae(seed=3))
    ref_keys = {"encoder." + k for k in O.make_vae_state_dict(CHAIN_VAE, "x")}  # the generator is strict-loaded into the reference
    path = str(tmp_path / "ckpt" / "checkpoint_last.pt")
    state = checkpoint.save_checkpoint(path, model, args=types.SimpleNamespace(arch="speech_vae_decoder", latent_dim=32), num_updates=7)
    assert set(state) >= {"args", "cfg", "model", "criterion", "optimizer_history", "task_state", "extra_state"}
    assert set(state["model"]) == ref_keys
    assert state["optimizer_history"][-1]["num_updates"] == 7
    other = FairseqEncoderModel(_vae(seed=4))
    assert not torch.equal(other.state_dict()["encoder.decoder_lm.weight"], model.state_dict()["encoder.decoder_lm.weight"])
    loaded = checkpoint.load_checkpoint(path, other)
    # top-level entries as the reference trainer writes them: "args" None, "cfg" nested (checkpoint_utils.py:423-426 takes the cfg branch)
    assert loaded["args"] is None and loaded["cfg"]["model"]["arch"] == "speech_vae_decoder" and loaded["cfg"]["model"]["_name"] == "speech_vae_decoder"
    assert {"model", "task", "criterion", "common"} <= set(loaded["cfg"])
    for k, v in model.state_dict().items():
        assert torch.equal(other.state_dict()[k], v), k
    # shapes are the reference's: conv [Cout, Cin, k], linear [out, in]
    ref_sd = O.make_vae_state_dict(CHAIN_VAE, "x")
    for k, v in ref_sd.items():
        assert state["model"]["encoder." + k].shape == v.shape, k


def test_plugin_loads_the_vae_checkpoint_for_the_diffusion_model(tmp_path):
    from diffnorm_amd import checkpoint
    from diffnorm_amd.fairseq_plugin.models import diff_discrete
    from diffnorm_amd.fairseq_plugin.registry import FairseqEncoderModel

    src = FairseqEncoderModel(_vae(seed=5))
    path = str(tmp_path / "vae.pt")
    checkpoint.save_checkpoint(path, src)
    args = types.SimpleNamespace(speech_decoder_ckpt=path, latent_dim=CHAIN_VAE.latent_dim, feature_dim=CHAIN_VAE.dim, hip_dtype="f32",
                                 denoiser_dim=CHAIN_EPS.dim, multitask=True, diffusion_timesteps=200)
    dec = diff_discrete._load_speech_decoder(args)
    for k, v in src.encoder.state_dict().items():
        assert torch.equal(dec.encoder.state_dict()[k], v), k
    # the diffusion model built on it: frozen VAE under encoder.speech_decoder.*, denoiser under encoder.model.*
    model = diff_discrete.DiffDiscreteModel.build_model(args, task=None)
    sd = model.state_dict()
    vae_keys = {k for k in sd if k.startswith("encoder.speech_decoder.")}
    eps_keys = {k for k in sd if k.startswith("encoder.model.")}
    assert len(vae_keys) == len(src.encoder.state_dict()) and vae_keys | eps_keys == set(sd)
    want_eps = {"encoder.model." + k for k in O.make_eps_state_dict(CHAIN_EPS, "x")} | {"encoder.model.pos_embed._float_tensor"}
    assert eps_keys == want_eps
    assert all(not p.requires_grad for n, p in model.named_parameters() if n.startswith("encoder.speech_decoder."))
    ck = str(tmp_path / "diff.pt")
    checkpoint.save_checkpoint(ck, model)
    fresh = diff_discrete.DiffDiscreteModel.build_model(args, task=None)
    checkpoint.load_checkpoint(ck, fresh)
    for k, v in sd.items():
        assert torch.equal(fresh.state_dict()[k], v), k


def test_cfg_holds_only_what_omegaconf_can_hold():
    """The reference's load path runs OmegaConf.create(state["cfg"]) (fairseq/checkpoint_utils.py): every value of the nested cfg must
    be a primitive, a list or a dict of those -- a namespace that carries a device, a dtype or a callable is written as strings
    (ADVICE round 3)."""
    from diffnorm_amd import checkpoint

    args = types.SimpleNamespace(arch="diff_discrete", lr=[1e-4], adam_betas="(0.9, 0.98)", device=torch.device("cpu"), dtype=torch.float16,
                                 hook=len, nested={"a": torch.device("cpu"), 3: (1, 2.5, None)}, max_tokens=15000, task="speech_diffusion_discrete",
                                 tensor=torch.zeros(2))
    cfg = checkpoint.nested_cfg(args)

    def ok(v):
        if v is None or isinstance(v, (bool, int, float, str)):
            return True
        if isinstance(v, list):
            return all(ok(x) for x in v)
        if isinstance(v, dict):
            return all(isinstance(k, str) and ok(x) for k, x in v.items())
        return False

    assert ok(cfg), cfg
    assert cfg["model"]["device"] == "cpu" and cfg["model"]["dtype"] == "torch.float16" and cfg["model"]["lr"] == [1e-4]
    assert cfg["model"]["nested"] == {"a": "cpu", "3": [1, 2.5, None]} and cfg["dataset"]["max_tokens"] == 15000 and cfg["task"]["_name"] == "speech_diffusion_discrete"
