"""Checkpoints in the reference's layout (fairseq/checkpoint_utils.py:35-186 writes what fairseq/trainer.py:392-436 assembles;
load_model_ensemble_and_task :391-493 reads `state["model"]` into the model built from `state["cfg"]/["args"]`).

A file written here is a `torch.save`d dict with the trainer's keys -- "args", "cfg", "model", "criterion", "optimizer_history",
"task_state", "extra_state", "last_optimizer_state" -- whose "model" entry holds the tensors under the reference's names and
shapes (`encoder.*` for the VAE, 138 tensors; `encoder.model.*` + `encoder.speech_decoder.*` for the diffusion
model, 516 tensors), whatever layout the HIP engines keep them in: the reference can `load_state_dict(strict=True)` it, and a
checkpoint written by the reference loads here.  "args" is None and "cfg" the nested {model, task, criterion, common, ...}
container, as the reference trainer writes them; the optimizer state is native-only (see save_checkpoint).
"""
import os
from typing import Any, Dict, Optional

import torch


def model_state(model) -> Dict[str, torch.Tensor]:
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


_CFG_GROUPS = {  # which namespace attributes the reference's nested config keeps where (fairseq/dataclass/configs.py): enough for
    # load_model_ensemble_and_task's `cfg.model` / `cfg.task` / `cfg.criterion` / `cfg.common` lookups (checkpoint_utils.py:423-470)
    "task": ("task", "data", "config_yaml", "max_source_positions", "max_target_positions", "target_is_code", "target_code_size",
             "n_frames_per_step", "src_feat_dir", "tgt_feat_dir"),
    "criterion": ("criterion",),
    "common": ("seed", "cpu", "fp16", "bf16", "amp", "user_dir"),
    "optimizer": ("optimizer", "adam_betas", "adam_eps", "weight_decay"),
    "lr_scheduler": ("lr_scheduler", "warmup_updates", "warmup_init_lr"),
    "optimization": ("lr", "clip_norm", "max_update", "update_freq"),
    "dataset": ("max_tokens", "batch_size", "train_subset", "valid_subset"),
}


def _config_value(v):
    """What OmegaConf.create can hold (the reference's load path runs `OmegaConf.create(state["cfg"])`, fairseq/checkpoint_utils.py:
    convert_namespace / load_checkpoint_to_cpu): None, bool, int, float, str and lists / dicts of those.  Anything else a namespace may
    carry (a device, a dtype, a callable, a tensor) is written as its string."""
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if isinstance(v, (list, tuple)):
        return [_config_value(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _config_value(x) for k, x in v.items()}
    return str(v)


def nested_cfg(args) -> Optional[Dict[str, Dict[str, Any]]]:
    """The flat namespace as the nested container the reference trainer writes under "cfg" (trainer.state_dict, fairseq/trainer.py:
    392-436: {model, task, criterion, common, ...}, a plain-dict rendering of its OmegaConf object); `model` carries every attribute
    (the reference's `cfg.model` is the whole argparse namespace for registry-built models, `_name` = the arch)."""
    if args is None:
        return None
    flat = dict(args) if isinstance(args, dict) else dict(vars(args))
    flat = {k: _config_value(v) for k, v in flat.items()}
    cfg = {"model": dict(flat, _name=flat.get("arch"))}
    for group, keys in _CFG_GROUPS.items():
        cfg[group] = {k: flat[k] for k in keys if k in flat}
        name = flat.get(group if group not in ("optimization", "dataset", "common") else "")
        if name is not None:
            cfg[group]["_name"] = name
    return cfg


def save_checkpoint(path: str, model, args=None, criterion=None, optimizer=None, lr_scheduler_state: Optional[dict] = None,
                    num_updates: int = 0, extra_state: Optional[dict] = None) -> Dict[str, Any]:
    """The optimizer entry: the HIP-backed FlatOptimizer's moments are flat buffers in the PACKED layout -- resumable here, not by the
    reference (whose Adam keeps per-parameter exp_avg / exp_avg_sq).  They are stored under "last_optimizer_state" with
    "optimizer_name" = "FlatOptimizer": the reference's trainer ASSERTS that the stored optimizer class equals its own
    (trainer.py:520-533) and raises otherwise, so resuming such a checkpoint in the reference needs `--reset-optimizer` (which skips
    the optimizer state); the MODEL entry is what is interchangeable both ways."""
    state = {
        "args": None,  # as the reference trainer writes it (legacy slot; load_model_ensemble_and_task takes the "cfg" branch then)
        "cfg": nested_cfg(args),
        "model": model_state(model),
        "criterion": None,
        "optimizer_history": [{
            "criterion_name": criterion.__class__.__name__ if criterion is not None else None,
            "optimizer_name": optimizer.__class__.__name__ if optimizer is not None else None,
            "lr_scheduler_state": lr_scheduler_state or {},
            "num_updates": int(num_updates),
        }],
        "task_state": {},
        "extra_state": dict(extra_state or {}),
    }
    if optimizer is not None:
        state["last_optimizer_state"] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)  # atomic, like checkpoint_utils.torch_persistent_save's rename
    return state


def load_checkpoint(path: str, model=None, optimizer=None, strict: bool = True) -> Dict[str, Any]:
    state = torch.load(path, map_location="cpu", weights_only=False)
    if model is not None:
        model.load_state_dict(state["model"], strict=strict)
    if optimizer is not None and state.get("last_optimizer_state") is not None:
        optimizer.load_state_dict(state["last_optimizer_state"])
    return state
