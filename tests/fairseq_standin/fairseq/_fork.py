"""What the DiffNorm fork has already registered when a --user-dir plugin is imported: the six names, under classes
whose NAMES equal the plugin's (speech_decoder_task.py:33, speech_diffusion_discrete_task.py:33,
speech_vae_decoder_loss.py:14, ddpm_discrete_loss.py:14, speech_vae_decoder.py:25,101, diff_discrete.py:25,137)."""
from .criterions import FairseqCriterion, register_criterion
from .models import FairseqEncoderModel, register_model, register_model_architecture
from .tasks import FairseqTask, register_task


@register_task("speech_decoder")
class SpeechDecoderTask(FairseqTask):
    origin = "fork"


@register_task("speech_diffusion_discrete")
class SpeechDiffusionDiscreteTask(FairseqTask):
    origin = "fork"


@register_criterion("speech_vae_decoder_loss")
class SpeechVAEDecoderLoss(FairseqCriterion):
    origin = "fork"


@register_criterion("ddpm_discrete_loss")
class DDPMDiscreteLoss(FairseqCriterion):
    origin = "fork"


@register_model("speech_vae_decoder")
class SpeechVAEDecoder(FairseqEncoderModel):
    origin = "fork"


@register_model_architecture("speech_vae_decoder", "speech_vae_decoder")
def _vae_arch(args):
    pass


@register_model("diff_discrete")
class DiffDiscreteModel(FairseqEncoderModel):
    origin = "fork"


@register_model_architecture("diff_discrete", "diff_discrete")
def _diff_arch(args):
    pass
