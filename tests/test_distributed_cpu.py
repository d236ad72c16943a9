"""CPU (-m "not gpu"): the N>1 path — flat gradient buckets + all-reduce + per-rank data shards —
exercised with world_size 2 over gloo on a small torch module (the HIP model itself needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hupr_amd.tools.distributed import GradientBuckets
    torch.manual_seed(1234 + rank)          # different init per rank on purpose: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 1))
    gb = GradientBuckets(net, bucket_bytes=1024)      # small buckets -> several collectives
    gb.broadcast_parameters(0)
    assert len(gb.buckets) >= 2
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)             # data-parallel shard of the global batch
    for _ in range(2):                                 # two iterations: views/zeroing are re-armed
        gb.prepare()
        loss = ((net(X[shard]) - Y[shard]) ** 2).sum()
        loss.backward()
        gb.finish()
    flat = torch.cat([b.flat_grad for b in gb.buckets])
    # every p.grad is still a view into its bucket
    for b in gb.buckets:
        for p, v in zip(b.params, b.views):
            assert p.grad.data_ptr() == v.data_ptr()
    out[rank] = (flat.clone(), torch.cat([b.flat_param for b in gb.buckets]).clone())
    if rank == 0:
        # single-process reference on the full batch with the broadcast weights
        ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 1))
        ref.load_state_dict(net.state_dict())
        ((ref(X) - Y) ** 2).sum().backward()
        want = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
        out["ref"] = want
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, p0 = out[0]
    g1, p1 = out[1]
    assert torch.equal(p0, p1), "parameters must be identical after broadcast"
    assert torch.allclose(g0, g1) and torch.allclose(g0, out["ref"], atol=1e-5), "sum of shard grads == full-batch grad"


def test_single_process_buckets_are_flat_views():
    from hupr_amd.tools.distributed import GradientBuckets
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
    before = [p.detach().clone() for p in net.parameters()]
    gb = GradientBuckets(net, bucket_bytes=1 << 20)
    assert len(gb.buckets) == 1 and gb.world_size == 1
    for p, q in zip(net.parameters(), before):
        assert torch.equal(p, q)
    net(torch.ones(3, 4)).sum().backward()
    gb.finish()
    b = gb.buckets[0]
    assert b.flat_grad.abs().sum() > 0 and b.pending == 0
    # parameters alias the flat buffer: an in-place flat update is visible through the module
    b.flat_param.zero_()
    assert all(p.abs().sum() == 0 for p in net.parameters())


def test_tail_bucket_is_small_and_covers_every_parameter_once():
    """The trailing parameters (the last gradients of backward) get a bucket of their own of at most tail_bytes: its
    all-reduce is the only one that cannot overlap with backward.  Every parameter sits in exactly one bucket, in
    reverse registration order."""
    from hupr_amd.tools.distributed import GradientBuckets
    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(12)])          # 12 x (16 KiB + 256 B)
    gb = GradientBuckets(net, bucket_bytes=60 << 10, tail_bytes=20 << 10)
    sizes = [b.numel * 4 for b in gb.buckets]
    assert sizes[-1] <= 20 << 10 and len(gb.buckets) >= 3, sizes
    order = [id(p) for b in gb.buckets for p in b.params]
    assert order == [id(p) for p in reversed(list(net.parameters()))]
    # tail_bytes = 0 switches the split off
    gb0 = GradientBuckets(torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(12)]), bucket_bytes=60 << 10, tail_bytes=0)
    assert len(gb0.buckets) == len(gb.buckets) - 1


def _accum_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hupr_amd.tools.distributed import GradientBuckets
    torch.manual_seed(99)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 1))
    gb = GradientBuckets(net, bucket_bytes=1024)
    g = torch.Generator().manual_seed(3)
    X, Y = torch.randn(24, 16, generator=g), torch.randn(24, 1, generator=g)
    for _ in range(2):                                  # two optimiser steps: the accumulator must be re-armed
        for m in range(3):                              # 3 micro-batches of 4 samples per rank per optimiser step
            lo = (rank * 3 + m) * 4
            gb.prepare(reduce=m == 2)
            ((net(X[lo:lo + 4]) - Y[lo:lo + 4]) ** 2).sum().backward()
            if m < 2:
                gb.stash()
        gb.finish()
    out[rank] = torch.cat([b.flat_grad for b in gb.buckets]).clone()
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 1))
        ref.load_state_dict(net.state_dict())
        ((ref(X) - Y) ** 2).sum().backward()
        out["ref"] = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
    dist.barrier()
    dist.destroy_process_group()


def test_accumulated_micro_batches_world2_one_exchange_per_step():
    """bench.py --strong: m micro-batches per rank, gradients summed locally, ONE all-reduce per optimiser step."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_accum_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.allclose(out[0], out[1]) and torch.allclose(out[0], out["ref"], atol=1e-5)


def test_single_rank_accumulation_sums_micro_batches():
    from hupr_amd.tools.distributed import GradientBuckets
    torch.manual_seed(5)
    net = torch.nn.Linear(6, 3)
    gb = GradientBuckets(net)
    X = torch.randn(8, 6)
    for m in range(2):
        gb.prepare(reduce=m == 1)
        net(X[m * 4:m * 4 + 4]).sum().backward()
        if m == 0:
            gb.stash()
    gb.finish()
    ref = torch.nn.Linear(6, 3)
    ref.load_state_dict(net.state_dict())
    ref(X).sum().backward()
    want = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
    assert torch.allclose(gb.buckets[0].flat_grad, want, atol=1e-6)


def test_fused_adam_checkpoint_interchanges_with_torch_adam():
    """the flat-bucket moments must round-trip through torch.optim.Adam's per-parameter state_dict layout
    (reference tools/base.py:76-81 saves optimizer.state_dict(), :113 restores it)."""
    from hupr_amd.tools.distributed import GradientBuckets
    from hupr_amd.tools.optim import FusedAdam
    torch.manual_seed(11)

    def make():
        return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    net = make()
    gb = GradientBuckets(net, bucket_bytes=256, tail_bytes=0)
    assert len(gb.buckets) >= 2
    opt = FusedAdam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.attach_flat_buckets(gb.flat_pairs(), gb.layout())
    assert opt.state_dict()["state"] == {}                      # nothing before the first step, like torch
    for st in opt._flat_state:                                  # as if 7 steps had run
        st["step"] = 7
        st["exp_avg"].normal_()
        st["exp_avg_sq"].uniform_(0.1, 1.0)
    sd = opt.state_dict()
    assert sorted(sd["state"]) == list(range(6)) and sd["param_groups"][0]["params"] == list(range(6))
    # (a) a torch.optim.Adam over the same parameters accepts it and sees the right slices
    tad = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    tad.load_state_dict(sd)
    for i, entries in enumerate(gb.layout()):
        for p, off, n in entries:
            assert float(tad.state[p]["step"]) == 7
            assert torch.equal(tad.state[p]["exp_avg"].reshape(-1), opt._flat_state[i]["exp_avg"][off:off + n])
            assert torch.equal(tad.state[p]["exp_avg_sq"].reshape(-1), opt._flat_state[i]["exp_avg_sq"][off:off + n])
    # (b) a fresh FusedAdam restores the flat buffers from torch's state_dict
    net2 = make()
    gb2 = GradientBuckets(net2, bucket_bytes=256, tail_bytes=0)
    opt2 = FusedAdam(net2.parameters(), lr=5e-4, weight_decay=1e-4)
    opt2.attach_flat_buckets(gb2.flat_pairs(), gb2.layout())
    opt2.load_state_dict(tad.state_dict())
    assert opt2.param_groups[0]["lr"] == 1e-3 and len(opt2.state) == 0
    for a, b in zip(opt._flat_state, opt2._flat_state):
        assert b["step"] == 7 and torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])


def _uid_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hupr_amd.tools.distributed import RcclTransport
    # two communicators in a row: the second exchange must not read the first one's key
    a = RcclTransport._exchange_id(bytes([rank + 1] * 128) if rank == 0 else None, None, torch.device("cpu"))
    b = RcclTransport._exchange_id(bytes([7] * 128) if rank == 0 else None, None, torch.device("cpu"))
    out[rank] = (a, b)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_id_exchange_through_the_store_world2():
    """The only multi-rank-specific step of the native RCCL transport that cannot run on a 1-GPU box: every rank must end up
    with rank 0's 128-byte communicator id (hupr_comm_unique_id), via the rendezvous store, once per communicator."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_uid_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] == (bytes([1] * 128), bytes([7] * 128))


def _agree_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hupr_amd.tools import distributed as D
    cpu = torch.device("cpu")
    res = [D._all_ok(True, None, cpu), D._all_ok(rank == 0, None, cpu), D._all_ok(rank == 1, None, cpu), D._all_ok(False, None, cpu)]
    # rank 0 could not create an id: it publishes the sentinel, the waiting rank gets it instead of blocking forever
    got = D.RcclTransport._exchange_id(D._ID_ERROR if rank == 0 else None, None, cpu)
    out[rank] = (res, got == D._ID_ERROR)
    dist.barrier()
    dist.destroy_process_group()


def test_transport_choice_is_a_group_decision_world2():
    """a native-communicator failure on SOME ranks must not split the job over two transports or leave ranks
    blocked: every phase of RcclTransport's construction is agreed with a MIN all-reduce (all ranks see the same verdict),
    and rank 0 publishes an error sentinel under the store key instead of leaving the other ranks waiting for an id."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_agree_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] == ([True, False, False, False], True)


def _no_cpu_backend_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hupr_amd.tools import distributed as D
    # a group WITHOUT a CPU backend (plain init_process_group("nccl")), imitated on gloo: collectives on ordinary CPU tensors
    # raise; tensors that went through ``.to(the transport's device)`` — tagged here, the stand-in device being the CPU — pass
    real_all_reduce, real_to = dist.all_reduce, torch.Tensor.to
    moved, seen = set(), []

    class _Dev:                                              # what RcclTransport passes as its device
        type = "cuda"

    def fake_all_reduce(t, *a, **k):
        if id(t) not in moved:
            raise RuntimeError("No backend type associated with device type cpu")
        return real_all_reduce(t, *a, **k)

    def fake_to(self, dev, *a, **k):
        if isinstance(dev, _Dev):
            seen.append("moved to the transport's device")
            r = self.clone()
            moved.add(id(r))
            return r
        return real_to(self, dev, *a, **k)

    dist.all_reduce, torch.Tensor.to = fake_all_reduce, fake_to
    try:
        got = D.RcclTransport._exchange_id(bytes([9] * 128) if rank == 0 else None, None, _Dev())
    finally:
        dist.all_reduce, torch.Tensor.to = real_all_reduce, real_to
    out[rank] = (got, list(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_id_exchange_on_a_group_without_a_cpu_backend_world2():
    """on a group with no CPU backend the agreement flag of the id exchange must fall back to the
    transport's GPU — it used to be handed torch.device("cpu"), so the fallback was a no-op, RcclTransport failed on every
    rank, make_transport fell back to torch.distributed and TrainEngine.capture() then refused to run.  An nccl-only group
    cannot exist on this CPU box; its behaviour (CPU collectives raise) is imitated on gloo and the exchange must still
    deliver rank 0's id, having moved its flag to the device it was given."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_no_cpu_backend_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][0] == out[1][0] == bytes([9] * 128)
    assert out[0][1] == out[1][1] == ["moved to the transport's device"]


class _SinkLinear(torch.autograd.Function):
    """y = x W^T + b whose backward writes dW / db straight into the gradient sink's views and returns None for them — the protocol
    of hupr_amd.functional's operators (``_pgrad`` / ``_pret``), with torch arithmetic standing in for the kernels."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        from hupr_amd import functional as F_
        x, w, b = ctx.saved_tensors
        dw, dw_direct = F_._pgrad(w)
        db, db_direct = F_._pgrad(b)
        dw.copy_(dy.t() @ x)                       # "the kernel": overwrites its slot
        db.copy_(dy.sum(0))
        return dy @ w, F_._pret(w, dw, dw_direct), F_._pret(b, db, db_direct)


class _SinkNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b, self.c = torch.nn.Linear(16, 32), torch.nn.Linear(32, 8), torch.nn.Linear(8, 1)

    def forward(self, x):
        x = torch.relu(_SinkLinear.apply(x, self.a.weight, self.a.bias))
        x = _SinkLinear.apply(x, self.b.weight, self.b.bias)
        return self.c(x)                           # the last layer through plain autograd: both gradient paths in one bucket set


def _sink_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hupr_amd import functional as F_
    from hupr_amd.tools.distributed import GradientBuckets
    torch.manual_seed(99 + rank)
    net = _SinkNet()
    gb = GradientBuckets(net, bucket_bytes=1 << 20, tail_bytes=0)     # ONE bucket: an early launch would miss its late gradients
    gb.direct = True                                                  # (the sink is a GPU feature; its host logic is device-agnostic)
    gb.broadcast_parameters(0)
    launches = []
    orig = gb._launch

    def launch(b):
        launches.append((sum(b.arrived), len(b.params)))
        return orig(b)
    gb._launch = launch
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 1, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)
    for it in range(3):
        gb.prepare()
        ((net(X[shard]) - Y[shard]) ** 2).sum().backward()
        gb.finish()
        assert F_.GRAD_SINK is None
    flat = torch.cat([b.flat_grad for b in gb.buckets]).clone()
    out[rank] = (flat, launches, [b.pending for b in gb.buckets], [b.zeroed for b in gb.buckets])
    if rank == 0:
        ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 1))
        ref.load_state_dict({k.replace("a.", "0.").replace("b.", "2.").replace("c.", "3."): v for k, v in net.state_dict().items()})
        ((ref(X) - Y) ** 2).sum().backward()
        out["ref"] = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))])
    dist.barrier()
    dist.destroy_process_group()


def test_direct_gradient_sink_world2_launches_after_the_last_gradient():
    """Round 5: operators that write parameter gradients straight into the bucket views return None for them, yet the autograd engine
    still visits those parameters' accumulation nodes (and runs the post-accumulate hooks).  Rounds 1-4 counted such a parameter
    twice and would have launched a bucket's all-reduce half-way through its gradients.  World size 2 over gloo, one bucket holding
    four kernel-written and two autograd-accumulated gradients: every launch sees all six arrived, one launch per pass, the pass ends
    at pending == 0, and the summed gradients equal the single-process full-batch gradient.  (Mixed buckets are zero-filled every
    pass: the un-zeroed form needs every slot kernel-written.)"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sink_worker, args=(world, port, out), nprocs=world, join=True)
    (g0, l0, p0, z0), (g1, l1, p1, z1) = out[0], out[1]
    assert l0 == [(6, 6)] * 3 and l1 == [(6, 6)] * 3, (l0, l1)
    assert p0 == [0] and p1 == [0] and z0 == [True] and z1 == [True]
    assert torch.allclose(g0, g1) and torch.allclose(g0, out["ref"], atol=1e-5)
