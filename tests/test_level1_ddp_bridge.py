"""CPU: the autograd bridge's two ways of delivering the flat gradient (diffnorm_amd/latent_module._finish_backward) and the
detection of torch's DistributedDataParallel around the model (fairseq's default --ddp-backend pytorch_ddp:
fairseq/dataclass/configs.py:301-309, fairseq/models/distributed_fairseq_model.py:59-84), with a recorder in place of the HIP
engine.  The GPU run with two real ranks is tests/test_hip_exchange.py::test_level1_under_torch_ddp."""
import types

import torch

from diffnorm_amd import latent_module as LM


class _Eng:
    def __init__(self, n):
        self.grads, self.work_current, self.synced = torch.zeros(n), True, 0

    def sync_work(self):
        self.synced += 1
        self.work_current = True

    def zero_grad(self):
        self.grads.zero_()

    def backward(self):
        self.grads += torch.arange(1.0, self.grads.numel() + 1)  # the engine ADDS d loss / d theta


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, eng, holder):
        ctx.fresh = LM._prepare_step(flat, eng, holder)
        ctx.flat, ctx.eng, ctx.holder = flat, eng, holder
        return flat.detach().sum().reshape(1) * 0 + 1.0

    @staticmethod
    def backward(ctx, g):
        return LM._finish_backward(ctx.flat, ctx.eng, ctx.holder, ctx.eng.backward, float(g), ctx.fresh), None, None


def _setup(through):
    eng = _Eng(4)
    flat = torch.nn.Parameter(torch.zeros(4))
    flat.grad = eng.grads
    return eng, flat, types.SimpleNamespace(_grads_through_autograd=through)


def test_in_place_form_keeps_the_alias_and_accumulates():
    eng, flat, holder = _setup(False)
    for k in (1, 2):
        (_Fn.apply(flat, eng, holder) * 2.0).backward()
        assert flat.grad is eng.grads and torch.equal(flat.grad, 2.0 * k * torch.arange(1.0, 5.0))
    flat.grad = None  # fairseq's zero_grad
    assert eng.work_current is False  # a backward ran: the next forward re-synchronises the working copies
    _Fn.apply(flat, eng, holder).backward()
    assert eng.synced == 2 and flat.grad is eng.grads and torch.equal(flat.grad, torch.arange(1.0, 5.0))


def test_through_autograd_form_feeds_the_accumulator():
    eng, flat, holder = _setup(True)
    fired = []
    flat.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad.clone()))  # where torch-DDP's reducer listens
    flat.grad = None
    (_Fn.apply(flat, eng, holder) * 3.0).backward()
    assert len(fired) == 1 and flat.grad is not eng.grads and torch.equal(flat.grad, 3.0 * torch.arange(1.0, 5.0))
    _Fn.apply(flat, eng, holder).backward()  # a second micro-batch accumulates in flat.grad, not in the engine's buffer
    assert len(fired) == 2 and torch.equal(flat.grad, 4.0 * torch.arange(1.0, 5.0)) and torch.equal(eng.grads, torch.arange(1.0, 5.0))
    (_Fn.apply(flat, eng, holder) * 0.0).backward()  # ignore_grad: zeros still reach the reducer (every rank feeds it every step)
    assert len(fired) == 3 and torch.equal(flat.grad, 4.0 * torch.arange(1.0, 5.0))


def test_an_alias_left_from_earlier_steps_is_detached_first():
    eng, flat, holder = _setup(True)
    eng.grads += 7.0  # accumulated in place before the wrapper appeared; flat.grad is eng.grads
    _Fn.apply(flat, eng, holder).backward()
    assert flat.grad is not eng.grads and torch.equal(flat.grad, 7.0 + torch.arange(1.0, 5.0))


def test_wrapper_detection():
    class Proxy(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m

    class FakeDDP(torch.nn.parallel.DistributedDataParallel):
        def __init__(self, m):  # no process group needed for an isinstance check
            torch.nn.Module.__init__(self)
            self.module = m

    lin = torch.nn.Linear(2, 2)
    assert not LM.wrapped_by_torch_ddp(lin) and not LM.wrapped_by_torch_ddp(Proxy(Proxy(lin)))
    assert LM.wrapped_by_torch_ddp(FakeDDP(lin)) and LM.wrapped_by_torch_ddp(Proxy(FakeDDP(lin)))
