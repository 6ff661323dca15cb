from . import speech_vae_decoder_loss, ddpm_discrete_loss  # noqa: F401
