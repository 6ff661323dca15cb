"""Profiler ranges with the reference's names.

The reference trainer brackets the phases of an update with `torch.autograd.profiler.record_function`: "forward" / "backward"
(fairseq/tasks/speech_decoder_task.py:215-220, fairseq/tasks/fairseq_task.py:512-520), "reduce-grads", "multiply-grads",
"clip-grads", "optimizer" (fairseq/trainer.py:912-958).  `profile_range(name)` opens the same torch range AND a roctx range
(libroctx64: what `rocprofv3 --marker-trace` records), so a trace of this build's update reads like a trace of the reference's.
Both are no-ops in cost when no profiler is attached (one C call each way).
"""
import contextlib
import ctypes

import torch

_roctx = None
_tried = False


def _lib():
    global _roctx, _tried
    if not _tried:
        _tried = True
        for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
    return _roctx


RANGES = ("forward", "backward", "reduce-grads", "multiply-grads", "clip-grads", "optimizer")


@contextlib.contextmanager
def profile_range(name: str):
    lib = _lib()
    if lib is not None:
        lib.roctxRangePushA(name.encode())
    try:
        with torch.autograd.profiler.record_function(name):
            yield
    finally:
        if lib is not None:
            lib.roctxRangePop()
