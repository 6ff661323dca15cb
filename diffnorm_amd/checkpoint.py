"""Checkpoints in the reference's layout (fairseq/checkpoint_utils.py:35-186 writes what fairseq/trainer.py:392-436 assembles;
load_model_ensemble_and_task :391-493 reads `state["model"]` into the model built from `state["cfg"]/["args"]`).

A file written here is a `torch.save`d dict with the trainer's keys -- "args", "cfg", "model", "criterion", "optimizer_history",
"task_state", "extra_state", "last_optimizer_state" -- whose "model" entry holds the tensors under the reference's names and
shapes (`encoder.*` for the VAE, 138 tensors; `encoder.model.*` + `encoder.speech_decoder.*` for the diffusion
model, 516 tensors), whatever layout the HIP engines keep them in: the reference can `load_state_dict(strict=True)` it, and a
checkpoint written by the reference loads here.
"""
import os
from typing import Any, Dict, Optional

import torch


def model_state(model) -> Dict[str, torch.Tensor]:
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def save_checkpoint(path: str, model, args=None, criterion=None, optimizer=None, lr_scheduler_state: Optional[dict] = None,
                    num_updates: int = 0, extra_state: Optional[dict] = None) -> Dict[str, Any]:
    state = {
        "args": args,  # legacy slot; fairseq >= 0.10 stores the config under "cfg"
        "cfg": vars(args) if (args is not None and not isinstance(args, dict)) else args,
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
