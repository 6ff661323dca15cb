"""One rank of tests/test_hip_exchange.py::test_level1_under_torch_ddp (started through torch.distributed.run; never imported by
pytest except for its helpers): the Level-1 path under fairseq's DEFAULT `--ddp-backend pytorch_ddp` -- the plugin model, built
from a training namespace and moved to the GPU, wrapped as fairseq/models/distributed_fairseq_model.py:59-84 wraps it (torch's
DistributedDataParallel inside a proxy that forwards attribute look-ups), handed to the plugin task's `train_step`, updated by an
optimizer that writes through `p.data` like fairseq's Adam.  Two ranks share cuda:0 over gloo (RCCL refuses two ranks on one
device).  Each rank trains on its own batch for three updates and writes what it ended with to <out>/rank<r>.npz."""
import math
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


class ModuleProxyWrapper(torch.nn.Module):
    """fairseq/distributed/module_proxy_wrapper.py restated: attribute look-ups that the wrapper does not have go to the twice
    wrapped module, forward goes to the DDP module."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            try:
                return getattr(self.module, name)
            except AttributeError:
                return getattr(self.module.module, name)

    def state_dict(self, *a, **k):
        return self.module.module.state_dict(*a, **k)

    def forward(self, *a, **k):
        return self.module(*a, **k)


class AdamThroughData:
    """fairseq/optim/adam.py:185-236 restated over the FairseqOptimizer slice a train step touches (zero_grad -> None)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.98), eps=1e-8):
        self._params, self.lr, self.betas, self.eps, self.state = [p for p in params if p.requires_grad], lr, betas, eps, {}

    @property
    def params(self):
        yield from self._params

    def backward(self, loss):
        loss.backward()

    def multiply_grads(self, c):
        for p in self._params:
            if p.grad is not None:
                p.grad.mul_(c)

    def zero_grad(self):
        for p in self._params:
            p.grad = None

    def step(self):
        for p in self._params:
            grad, p_data = p.grad.data.float(), p.data
            st = self.state.setdefault(p, {"step": 0, "m": torch.zeros_like(p_data), "v": torch.zeros_like(p_data)})
            st["step"] += 1
            b1, b2 = self.betas
            st["m"].mul_(b1).add_(grad, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(grad, grad, value=1 - b2)
            p_data.addcdiv_(st["m"], st["v"].sqrt().add_(self.eps), value=-self.lr * math.sqrt(1 - b2 ** st["step"]) / (1 - b1 ** st["step"]))


def build(dev):
    import diffnorm_oracle as O
    from diffnorm_amd import fairseq_plugin  # noqa: F401
    from diffnorm_amd.fairseq_plugin import registry
    from gen_golden_configs import CHAIN_VAE as CFG

    args = types.SimpleNamespace(arch="speech_vae_decoder", criterion="speech_vae_decoder_loss", latent_dim=CFG.latent_dim, feature_dim=CFG.dim,
                                 hip_dtype="f32", target_code_size=1000, data="", optimizer="adam", lr=[1e-3])
    task = registry.TASK_REGISTRY["speech_decoder"].setup_task(args)
    model = task.build_model(args)
    model.load_state_dict({"encoder." + k: v for k, v in O.make_vae_state_dict(CFG, "train").items()}, strict=True)
    model.to(dev)
    model.encoder.attn_dropout = 0.0
    return task, model, task.build_criterion(args), CFG


def make_sample(rank, dim, z, it):
    g = torch.Generator().manual_seed(1000 + rank)
    B, T = (2, 40) if rank == 0 else (3, 48)
    lens = torch.tensor([T, T - 13, T - 5][:B])
    mask = torch.arange(T).view(1, -1) < lens.view(-1, 1)
    feat = torch.randn(B, T, dim, generator=g) * mask.unsqueeze(-1)
    unit = torch.randint(4, 1004, (B, T), generator=g) * mask
    noise = torch.randn(B, T, z, generator=torch.Generator().manual_seed(77 + 10 * rank + it))
    return {"net_input": {"src_tokens": feat, "src_lengths": lens}, "reduce_target": feat, "reduce_target_unit": unit, "reduce_target_lengths": lens,
            "target": feat, "target_unit": unit, "target_lengths": lens, "ntokens": int(lens.sum()), "nsentences": B, "posterior_noise": noise}


def run(task, model, criterion, opt, ranks, world, n_updates, dim, z):
    """One process's share of `n_updates` updates as fairseq's trainer drives them (trainer.py:784-960): micro-batches -> (DDP
    averages over ranks) -> multiply_grads(world / sample_size) -> step.  ranks: the batches this process owns."""
    out = {}
    total_sentences = 2 + 3
    for it in range(n_updates):
        opt.zero_grad()
        for r in ranks:
            loss, n, _ = task.train_step(make_sample(r, dim, z, it), model, criterion, opt, it)
        opt.multiply_grads(world / total_sentences)
        if it == 0:
            out["grad0"] = next(iter(opt.params)).grad.detach().cpu().numpy().copy()
        opt.step()
        out[f"loss{it}"] = float(loss.detach())
    out["master"] = next(iter(opt.params)).detach().cpu().numpy()
    return out


def main():
    out_dir = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    task, model, criterion, cfg = build(dev)
    if rank == 1:  # DDP's constructor broadcasts rank 0's parameters: start rank 1 somewhere else to see that it arrives
        with torch.no_grad():
            model.encoder.flat_params.mul_(1.5)
    wrapped = ModuleProxyWrapper(torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, broadcast_buffers=False,
                                                                              bucket_cap_mb=25, find_unused_parameters=False))
    assert [n for n, _ in wrapped.named_parameters()] == ["module.module.encoder.flat_params"]
    opt = AdamThroughData(wrapped.parameters())
    res = run(task, wrapped, criterion, opt, [rank], world, 3, cfg.dim, cfg.z)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback

        with open(os.path.join(sys.argv[1], f"error_rank{os.environ.get('RANK', '0')}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise
