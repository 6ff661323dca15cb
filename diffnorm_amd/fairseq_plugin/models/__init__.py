from . import speech_vae_decoder, diff_discrete  # noqa: F401
