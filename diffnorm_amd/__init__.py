"""diffnorm_amd: MI355X-native (gfx950) implementation of DiffNorm's latent-diffusion denoising hot path.

Layout: csrc/ (HIP kernels + C-ABI engine -> libdiffnorm_hip.so), _lib.py (ctypes binding), ops.py
(op-level wrappers), packing.py (state-dict -> packed weights), engine.py (engine handles),
scheduler.py (noise schedules), latent_module.py (host mirror of the reference modules),
fairseq_plugin/ (register_model / register_task / register_criterion surface).
"""
__version__ = "0.1.0"
