"""The plugin's fairseq-present registration branch (diffnorm_amd/fairseq_plugin/registry.py), driven with the stand-in
`fairseq` package under tests/fairseq_standin (the decorator checks of the fork restated: duplicate name, duplicate CLASS
name, base class -- reference fairseq/tasks/__init__.py:48-101, fairseq/registry.py:62-100, fairseq/models/__init__.py:
109-207), in which the fork's six names are already registered under classes named like the plugin's.  Runs in a fresh
interpreter so the stand-in never leaks into the other tests."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "fairseq_standin")


def _run(code):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STANDIN, ROOT, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_standin_rejects_duplicates_like_the_fork():
    _run("""
        import fairseq
        from fairseq.tasks import register_task, FairseqTask
        from fairseq.criterions import register_criterion, FairseqCriterion
        class SpeechDecoderTask(FairseqTask): pass
        for deco, cls in ((register_task("speech_decoder"), SpeechDecoderTask), (register_task("other_name"), SpeechDecoderTask)):
            try:
                deco(cls)
            except ValueError as e:
                assert "duplicate" in str(e)
            else:
                raise SystemExit("duplicate accepted")
        class SpeechVAEDecoderLoss(FairseqCriterion): pass
        try:
            register_criterion("brand_new")(SpeechVAEDecoderLoss)
        except ValueError as e:
            assert "duplicate class name" in str(e)
        else:
            raise SystemExit("duplicate criterion class name accepted")
    """)


def test_plugin_replaces_the_forks_registrations_and_reloads():
    out = _run("""
        import importlib, sys
        import fairseq
        from fairseq import models as fm, tasks as ft, criterions as fc
        assert ft.TASK_REGISTRY["speech_decoder"].origin == "fork"
        import diffnorm_amd.fairseq_plugin as plug
        from diffnorm_amd.fairseq_plugin import registry
        assert registry.HAVE_FAIRSEQ
        def check():
            for table, names in ((ft.TASK_REGISTRY, ("speech_decoder", "speech_diffusion_discrete")),
                                 (fc.CRITERION_REGISTRY, ("speech_vae_decoder_loss", "ddpm_discrete_loss")),
                                 (fm.MODEL_REGISTRY, ("speech_vae_decoder", "diff_discrete")),
                                 (fm.ARCH_MODEL_REGISTRY, ("speech_vae_decoder", "diff_discrete", "speech_diffusion"))):
                for n in names:
                    cls = table[n]
                    assert cls.__module__.startswith("diffnorm_amd."), (n, cls.__module__)
                    assert not hasattr(cls, "origin")
            assert fm.ARCH_MODEL_INV_REGISTRY["diff_discrete"].count("diff_discrete") == 1
            assert issubclass(ft.TASK_REGISTRY["speech_decoder"], ft.FairseqTask)
            assert issubclass(fm.MODEL_REGISTRY["diff_discrete"], fm.BaseFairseqModel)
        check()
        # a second import of the plugin's modules (fairseq's --user-dir import followed by an explicit one) must not trip
        # the duplicate checks either
        for name in sorted(m for m in sys.modules if m.startswith("diffnorm_amd.fairseq_plugin.") and m.count(".") == 3):
            importlib.reload(sys.modules[name])
        check()
        print("ok")
    """)
    assert "ok" in out
