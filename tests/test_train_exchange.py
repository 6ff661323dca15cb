"""The exchange step of data-parallel training (training row): bucket plan, bucketed gradient all-reduce, the
statistics all-reduce and the trainer's gradient scaling, on CPU with two gloo ranks (world_size 2).  The engine and the
optimizer kernel are replaced by recorders: what is tested is the host logic around them (diffnorm_amd/training.py) --
every rank must end with the SUM of the ranks' gradients, scaled by 1 / (total sentences) like fairseq's
DDP-mean x multiply_grads(world / sample_size) (fairseq/trainer.py:912-933)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bucket_plan_merges_descending_ranges():
    from diffnorm_amd.training import plan_buckets

    # stage ranges as the VAE engine reports them: head, layers (top first), decoder WaveNets, encoder WaveNets
    ranges = [(900, 100), (700, 200), (500, 200), (300, 200), (100, 200), (0, 100)]
    assert plan_buckets(ranges, 1) == [(i, o, c) for i, (o, c) in enumerate(ranges)]
    b = plan_buckets(ranges, 250)
    assert b == [(1, 700, 300), (3, 300, 400), (5, 0, 300)]
    assert sum(c for _, _, c in b) == 1000 and b[-1][1] == 0
    assert plan_buckets(ranges, 10 ** 9) == [(5, 0, 1000)]
    with pytest.raises(AssertionError):
        plan_buckets([(0, 10), (20, 10)], 1)


class _FakeEngine:
    """Duck-types VaeTrainEngine on CPU: backward stage k writes (rank + 1) * (k + 1) into its range."""

    def __init__(self, rank):
        self.rank = rank
        self.device = torch.device("cpu")
        self.ranges = [(60, 40), (40, 20), (10, 30), (0, 10)]
        self.n_stages = len(self.ranges)
        self.master = torch.zeros(100)
        self.work = self.master
        self.grads = torch.zeros(100)
        self.order = []

    def stage_ranges(self):
        return self.ranges

    def zero_grad(self):
        self.grads.zero_()

    def forward(self, feat, units, lens, noise=None, ntokens=None):
        self.order.append("fwd")
        return torch.tensor([1.0 + self.rank, 2.0, 3.0, 4.0, 0.5, float(ntokens), 0.0, 0.0])

    def backward(self, first=0, last=None):
        last = self.n_stages - 1 if last is None else last
        for st in range(first, last + 1):
            off, cnt = self.ranges[st]
            self.grads[off: off + cnt] += (self.rank + 1.0) * (st + 1.0)
            self.order.append(f"bwd{st}")

    def refresh(self):
        self.order.append("refresh")


class _RecAdam:
    def __init__(self):
        self.calls = []

    def set_lr(self, lr):
        self.lr = lr

    def step(self, grads, grad_scale=1.0, grad_scale_dev=None):
        self.calls.append((grads.clone(), grad_scale, grad_scale_dev.clone()))
        return (grads.double().pow(2).sum().sqrt() * grad_scale * grad_scale_dev[0]).float()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffnorm_amd import training

    eng, adam = _FakeEngine(rank), _RecAdam()
    tr = training.VaeTrainer(eng, lr=1e-3, warmup_updates=10, warmup_init_lr=1e-7, adam=adam, bucket_mb=50 * 4 / (1 << 20))
    assert tr.reducer.world == world
    # two micro-batches on rank 0, one on rank 1 would desynchronise the collectives: fairseq gives every rank the same count
    nsent = [3, 5][rank]
    samples = [{"reduce_target": torch.zeros(nsent, 7, 4), "reduce_target_unit": torch.zeros(nsent, 7, dtype=torch.long),
                "reduce_target_lengths": torch.full((nsent,), 7), "ntokens": 7 * nsent, "nsentences": nsent} for _ in range(2)]
    logged, norm = tr.train_step(samples)
    torch.save({"grads": eng.grads.clone(), "adam_grads": adam.calls[0][0], "scale": adam.calls[0][1], "scale_dev": adam.calls[0][2],
                "logged": logged, "order": eng.order, "buckets": tr.reducer.buckets, "lr": adam.lr, "norm": norm}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gradient_exchange_and_scaling(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"r{k}.pt") for k in range(2)]
    # per stage st the two micro-batches of rank k wrote 2 * (k + 1) * (st + 1); the exchange sums the ranks: 2 * 3 * (st + 1)
    want = torch.zeros(100)
    for st, (off, cnt) in enumerate([(60, 40), (40, 20), (10, 30), (0, 10)]):
        want[off: off + cnt] = 2 * 3.0 * (st + 1)
    for k in range(2):
        assert torch.equal(r[k]["grads"], want) and torch.equal(r[k]["adam_grads"], want)
        assert r[k]["scale"] == 1.0
        assert abs(float(r[k]["scale_dev"]) - 1.0 / (2 * 3 + 2 * 5)) < 1e-9  # 1 / total sentences of the update
        assert r[k]["buckets"] == [(1, 40, 60), (3, 0, 40)]  # 50-element buckets over ranges of 40 / 20 / 30 / 10
        # first micro-batch: one backward call; last: stage by stage (buckets enter the all-reduce as they complete)
        assert r[k]["order"] == ["fwd", "bwd0", "bwd1", "bwd2", "bwd3", "fwd", "bwd0", "bwd1", "bwd2", "bwd3", "refresh"]
        assert abs(r[k]["lr"] - 1e-7) < 1e-15  # inverse_sqrt at num_updates = 0 is warmup_init_lr
    # logged statistics: sample-size-weighted means over both ranks (reduce_metrics); loss was 1 + rank
    tot = 2 * 3 + 2 * 5
    assert abs(float(r[0]["logged"][0]) - (2 * 3 * 1.0 + 2 * 5 * 2.0) / tot) < 1e-6
    assert torch.equal(r[0]["logged"], r[1]["logged"]) and float(r[0]["norm"]) == float(r[1]["norm"])
