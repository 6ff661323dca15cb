from . import speech_decoder_task, speech_diffusion_discrete_task  # noqa: F401
