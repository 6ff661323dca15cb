"""`create_diffusion` with the reference's signature and defaults (reference diffusion/__init__.py:10-46):
linear betas 1e-4..2e-2, 1000 steps, eps-prediction, learn_sigma=True -> LEARNED_RANGE, MSE loss."""
from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False, predict_xstart=False,
                     learn_sigma=True, rescale_learned_sigmas=False, diffusion_steps=1000):
    betas = gd.get_named_beta_schedule(noise_schedule, diffusion_steps)
    loss_type = gd.LossType.RESCALED_KL if use_kl else (gd.LossType.RESCALED_MSE if rescale_learned_sigmas else gd.LossType.MSE)
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    var = gd.ModelVarType.LEARNED_RANGE if learn_sigma else (gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE)
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, timestep_respacing), betas=betas,
                           model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
                           model_var_type=var, loss_type=loss_type)
