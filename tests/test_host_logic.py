"""CPU tests of the host-side logic around the path: plugin registration contract, schedules of the product side
against the golden tables, unit de-duplication / batch assembly / TSV lines, and the multi-rank sharding over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plugin_registers_reference_names():
    from diffnorm_amd.fairseq_plugin import registry as R
    import diffnorm_amd.fairseq_plugin  # noqa: F401

    assert {"speech_vae_decoder", "diff_discrete"} <= set(R.MODEL_REGISTRY)
    assert {"speech_vae_decoder", "diff_discrete", "speech_diffusion"} <= set(R.ARCH_MODEL_REGISTRY)
    assert {"speech_decoder", "speech_diffusion_discrete"} <= set(R.TASK_REGISTRY)
    assert {"speech_vae_decoder_loss", "ddpm_discrete_loss"} <= set(R.CRITERION_REGISTRY)
    if not R.HAVE_FAIRSEQ:  # the fallback keeps fairseq's decorator contract
        with pytest.raises(ValueError):
            R.register_model("diff_discrete")(R.MODEL_REGISTRY["diff_discrete"])
        with pytest.raises(ValueError):
            R.register_model("not_a_model")(object)
        with pytest.raises(ValueError):
            R.register_model_architecture("unknown_model", "x")(lambda a: None)


def test_plugin_cli_flags_and_task_setup():
    import argparse

    from diffnorm_amd.fairseq_plugin import registry as R
    import diffnorm_amd.fairseq_plugin  # noqa: F401

    p = argparse.ArgumentParser()
    R.MODEL_REGISTRY["diff_discrete"].add_args(p)
    args = p.parse_args(["--latent_dim", "128", "--speech_decoder_ckpt", "/x/ckpt.pt", "--encoder-embed-dim", "512"])
    assert args.latent_dim == 128 and args.speech_decoder_ckpt == "/x/ckpt.pt" and args.classifier_guidance == 1.0
    R.ARCH_CONFIG_REGISTRY["diff_discrete"](args)
    assert args.decoder_layers == 6 and args.decoder_embed_dim == 512
    tp = argparse.ArgumentParser()
    R.TASK_REGISTRY["speech_diffusion_discrete"].add_args(tp)
    targs = tp.parse_args(["/data", "--target-is-code", "--target-code-size", "1000", "--max-target-positions", "2048"])
    task = R.TASK_REGISTRY["speech_diffusion_discrete"].setup_task(targs)
    assert len(task.target_dictionary) == 1004 and task.target_dictionary.index("0") == 4
    ds = task.load_dataset("train", n=5, min_len=8, max_len=20)
    batch = ds.collater([ds[i] for i in range(3)])
    assert set(batch) >= {"net_input", "target", "target_unit", "reduce_target", "reduce_target_unit", "target_lengths",
                          "reduce_target_lengths", "ntokens", "nsentences", "id"}
    assert batch["reduce_target_unit"].min() >= 0 and batch["ntokens"] == int(batch["target_lengths"].sum())
    agg = R.CRITERION_REGISTRY["ddpm_discrete_loss"].reduce_metrics(
        [{"loss": 1.0, "noise_loss": 1.0, "mse_loss": 0.0, "nll_loss": 2.0, "acc": 0.5, "sample_size": 1},
         {"loss": 3.0, "noise_loss": 1.0, "mse_loss": 0.0, "nll_loss": 2.0, "acc": 0.5, "sample_size": 3}])
    assert abs(agg["loss"] - 2.5) < 1e-6 and agg["sample_size"] == 4


def test_product_schedules_match_reference_golden(golden):
    from diffnorm_amd import scheduler

    g = golden("schedules")
    for n in (200, 1000):
        s = scheduler.DDPMScheduler(n)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                  "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            np.testing.assert_allclose(getattr(s, k), g[f"ddpm{n}_{k}"], rtol=1e-14, atol=0)
        coef = s.ddim_coef_table()
        t = torch.tensor([0, 1, n // 2, n - 1])
        assert torch.equal(coef[t, 0], s.get_sqrt_alpha_cum(t, (4,)).view(-1))
        abp = s.get_alpha_prev_cum(t, (4,)).view(-1)
        assert torch.equal(coef[t, 2], torch.sqrt(abp)) and torch.equal(coef[t, 3], torch.sqrt(1 - abp))
    lin = scheduler.ScheduleTables(scheduler.get_named_beta_schedule("linear", 1000))
    np.testing.assert_allclose(lin.sqrt_recipm1_alphas_cumprod, g["linear1000_sqrt_recipm1_alphas_cumprod"], rtol=1e-14)


def test_reduce_token_and_batch_assembly():
    from diffnorm_amd import normalize as N

    dedup, dur, keep = N.reduce_token([5, 5, 7, 7, 7, 5, 9])
    assert dedup == [5, 7, 5, 9] and dur == [2, 3, 1, 1] and keep.tolist() == [0, 2, 5, 6]
    assert N.reduce_token([])[:2] == ([], []) and N.reduce_token([3])[0] == [3]
    g = torch.Generator().manual_seed(0)
    utts = []
    for i, units in enumerate(([1, 1, 2, 3, 3], [4, 5, 5])):
        feat = torch.randn(len(units), 768, generator=g)
        utts.append(N.Utterance(f"id{i}", f"a{i}.wav", 100 + i, feat, units, N.reduce_token(units)[0]))
    feat, unit, lens = N.assemble_batch(utts, "cpu")
    assert feat.shape == (2, 3, 768) and lens.tolist() == [3, 2] and unit.tolist() == [[1, 2, 3], [4, 5, 0]]
    assert torch.equal(feat[0, 1], utts[0].feat[2]) and feat[1, 2].abs().sum() == 0
    assert N.tsv_line(utts[1], [9, 9, 8]) == "id1\ta1.wav\t101\t9 8\t3"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from diffnorm_amd import normalize as N

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    utts = []
    for i in range(11):
        T = int(torch.randint(3, 9, (1,), generator=g))
        units = torch.randint(0, 50, (T,), generator=g).tolist()
        utts.append(N.Utterance(f"u{i}", f"s{i}.wav", T, torch.randn(T, 768, generator=g), units, N.reduce_token(units)[0]))

    def fake_ddim_sample(feat, input_mask=None, cond_scale=1.0, ref_units=None, start_step=50):
        # deterministic stand-in for the GPU sampler: a function of the features only
        pred = (feat.abs().sum(-1) * 7).long() % 50
        lens = input_mask.sum(1)
        return [pred[i, : int(lens[i])] for i in range(feat.shape[0])], 0, int(input_mask.sum()), None

    lines = N.normalize(fake_ddim_sample, utts, start_step=5, batch_size=3, device="cpu")
    q.put((rank, lines))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_normalisation_world2_matches_single_rank():
    """world_size 2 over gloo: every rank ends up with the same TSV lines, identical to the 1-rank run."""
    import torch.multiprocessing as mp

    from diffnorm_amd import sharding

    assert sharding.my_batches(5, 0, 2) == [0, 2, 4] and sharding.my_batches(5, 1, 2) == [1, 3]
    assert [list(r) for r in sharding.batch_indices(7, 3)] == [[0, 1, 2], [3, 4, 5], [6]]
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        port, q = _free_port(), ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=120) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        results[world] = dict(got)
    assert len(results[1][0]) == 11
    assert results[2][0] == results[2][1] == results[1][0]


def test_split_rows_round_trip_on_host():
    """split_rows / unsplit_rows: x = hi + lo to 2^-16 relative, the layout is 32 hi then 32 lo per group of 32."""
    from diffnorm_amd import packing

    def seeded(shape, seed):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))

    x = seeded((5, 96), 1) * torch.logspace(-6, 6, 96)
    s = packing.split_rows(x)
    assert s.dtype == torch.bfloat16 and s.shape == (5, 192)
    back = packing.unsplit_rows(s)
    assert ((back - x).abs() <= x.abs() * 2.0 ** -16).all()
    assert torch.equal(s[:, :32].float(), x[:, :32].to(torch.bfloat16).float())       # hi of group 0
    assert torch.equal(s[:, 64:96].float(), x[:, 32:64].to(torch.bfloat16).float())   # hi of group 1
    w = packing.split_rows(x, weight=True)                                              # weights: lo half first
    assert torch.equal(w[:, 32:64], s[:, :32]) and torch.equal(w[:, :32], s[:, 32:64])
    assert torch.equal(packing.unsplit_rows(w), back)


def test_profile_ranges_carry_the_reference_names():
    """diffnorm_amd.profiling: the phases of an update are bracketed with the names the reference trainer uses (fairseq/trainer.py:
    912-958, fairseq/tasks/speech_decoder_task.py:215-220) as torch profiler ranges (+ roctx where libroctx64 loads): a trace of this
    build's update reads like the reference's."""
    from diffnorm_amd import profiling

    assert profiling.RANGES == ("forward", "backward", "reduce-grads", "multiply-grads", "clip-grads", "optimizer")
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        for name in profiling.RANGES:
            with profiling.profile_range(name):
                torch.ones(4).sum()
    seen = {e.key for e in prof.key_averages()}
    assert set(profiling.RANGES) <= seen

