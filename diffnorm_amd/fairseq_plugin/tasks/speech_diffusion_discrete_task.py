"""`--task speech_diffusion_discrete` (reference fairseq/tasks/speech_diffusion_discrete_task.py:33): identical to
`speech_decoder` upstream except for the registered name/class."""
from ..registry import register_task
from .speech_decoder_task import _SpeechTaskBase


@register_task("speech_diffusion_discrete")
class SpeechDiffusionDiscreteTask(_SpeechTaskBase):
    pass
