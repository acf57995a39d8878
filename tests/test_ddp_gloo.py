"""Multi-process data parallelism on CPU (gloo, world_size 2): the bucketed overlapped reducer over the flat gradient
buffer gives every rank the mean gradient == the single-process gradient on the concatenated batch, parameters are
broadcast from rank 0, unused parameters do not hang the step, and the cross-replica BatchNorm statistics exchange
(pairs of per-channel sums) reproduces whole-batch statistics."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(12, 40)
        self.c = torch.nn.Conv2d(2, 3, 3, bias=False)
        self.b = torch.nn.Linear(40, 5)
        self.unused = torch.nn.Parameter(torch.ones(7))    # like CLIP.logit_scale: never receives a gradient
        self.register_buffer("running", torch.zeros(3))    # like a BatchNorm running statistic
        self.explicit_grad_ready = False
        self._store = None

    def prepare(self, device):
        from crog_amd.runtime import ParamStore
        self._store = ParamStore(self, torch.device(device))
        return self

    @property
    def store(self):
        return self._store

    def forward(self, x, img):
        return self.b(torch.tanh(self.a(x))).sum() + self.c(img).pow(2).mean()


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crog_amd.parallel import DistributedDataParallel, SyncBNComm, convert_sync_batchnorm
    from crog_amd.runtime import RT
    torch.manual_seed(100 + rank)                      # different init per rank: DDP must broadcast rank 0's weights
    net = Net().prepare("cpu")
    ddp = DistributedDataParallel(net, bucket_cap_mb=0.001)   # tiny buckets -> several collectives in flight
    g = torch.Generator().manual_seed(7)
    X, I = torch.randn(8, 12, generator=g), torch.randn(8, 2, 6, 6, generator=g)
    xs, im = X[rank * 4:(rank + 1) * 4], I[rank * 4:(rank + 1) * 4]
    # broadcast_buffers=True (torch DDP's default, train_crog.py:154-156) without SyncBatchNorm: rank 0's buffers win at every forward
    net.running.fill_(10.0 + rank)
    for step in range(2):
        net.store.zero_grad()
        loss = ddp(xs, im)
        loss.backward()
        ddp.reducer.wait()
    assert len(ddp.reducer.buckets) > 2
    assert torch.equal(net.running, torch.full((3,), 10.0)), (rank, net.running)
    # cross-replica BatchNorm statistics: all-reduce of (sum, sum^2) pairs
    convert_sync_batchnorm(net)         # installs the statistics communicator on a process group of its own
    comm = RT.comm
    assert isinstance(comm, SyncBNComm) and comm.group is not None and comm.group is not dist.group.WORLD and comm.world_size == world
    z = I[rank * 4:(rank + 1) * 4]
    pairs = torch.stack([z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))], 1).contiguous()
    comm.all_reduce_sum(pairs)
    assert comm.calls == 1 and ddp.reducer.launches == 2 * len(ddp.reducer.buckets)
    G32 = net.store.G.clone()
    # opt-in bf16 gradient payload: the same step with every bucket rounded for the wire
    ddp16 = DistributedDataParallel(net, bucket_cap_mb=0.001, gradient_payload=torch.bfloat16)
    net.store.zero_grad()
    ddp16(xs, im).backward()
    ddp16.reducer.wait()
    assert ddp16.reducer.payload_dtype is torch.bfloat16 and net.store.G.dtype == torch.float32
    torch.save(dict(G=G32, G16=net.store.G.clone(), P=net.store.P.clone(), pairs=pairs), os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_reducer_matches_single_process(tmp_path):
    world, port = 2, 29000 + os.getpid() % 2000
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["P"], r1["P"]) and torch.allclose(r0["G"], r1["G"], atol=1e-7)
    # single-process reference with rank 0's weights on the whole batch
    from crog_amd.runtime import ParamStore
    torch.manual_seed(100)
    net = Net()
    st = ParamStore(net, torch.device("cpu"))
    assert torch.equal(st.P, r0["P"])
    g = torch.Generator().manual_seed(7)
    X, I = torch.randn(8, 12, generator=g), torch.randn(8, 2, 6, 6, generator=g)
    # mean over ranks of per-rank losses == what DDP's averaged gradient corresponds to
    (0.5 * (net(X[:4], I[:4]) + net(X[4:], I[4:]))).backward()
    assert torch.allclose(st.G, r0["G"], atol=1e-6, rtol=1e-5)
    assert st.G[st.off(net.unused):st.off(net.unused) + 7].abs().sum() == 0
    # bf16 payload: identical on both ranks, equal to the fp32 mean up to the rounding of each rank's contribution (2^-8 of ITS size:
    # measured on the whole vector, the two contributions of an element may cancel)
    assert torch.equal(r0["G16"], r1["G16"])
    rel = float((r0["G16"] - r0["G"]).norm() / r0["G"].norm())
    assert 0 < rel < 2 ** -7, rel
    ref = torch.stack([I.sum((0, 2, 3)), (I * I).sum((0, 2, 3))], 1)
    assert torch.allclose(r0["pairs"], ref, atol=1e-4)


def _order_worker(rank, world, port, tmp):
    """Four ranks: every rank must issue the SAME gradient-bucket collectives in the SAME order (gloo, like RCCL, matches collectives of a
    communicator by issue order), each bucket exactly once per step - also when a weight gradient is still held back by the runtime at the
    end of backward (a deferred / parked launch: Runtime._pending_wgrad + the announcement waiting in _pending_done), which Reducer.wait must
    flush BEFORE it launches the remaining buckets (it ran ahead of the runtime's own end-of-backward flush, all-reduced the
    bucket without that gradient and was re-armed by the late announcement)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crog_amd import parallel as PP
    from crog_amd.runtime import RT
    torch.manual_seed(100)
    net = Net().prepare("cpu")
    net.explicit_grad_ready = True            # this test announces the gradients itself (as the crog_amd models do through WRef.done)
    ddp = DistributedDataParallel = PP.DistributedDataParallel(net, bucket_cap_mb=0.0005)
    order = []
    orig = PP.Reducer._launch

    def logged(self, b):
        if not b["launched"]:
            order.append(self.buckets.index(b))
        return orig(self, b)
    PP.Reducer._launch = logged
    parked = net.a.weight                      # its "GEMM" is enqueued late: the gradient arrives when the runtime flushes
    late = []

    def hook(p):
        if p is parked:
            g = p.grad.clone()
            p.grad.zero_()                     # not written yet ...
            RT._pending_wgrad.append((lambda: p.grad.add_(g), ()))      # ... the deferred launch writes it
            RT._pending_done.append(lambda: (late.append(len(order)), RT.reducer.mark_ready(p)))
        else:
            RT.reducer.mark_ready(p)
    for q in net.parameters():
        q.register_post_accumulate_grad_hook(hook)
    g = torch.Generator().manual_seed(7)
    X, I = torch.randn(4 * world, 12, generator=g), torch.randn(4 * world, 2, 6, 6, generator=g)
    xs, im = X[rank * 4:(rank + 1) * 4], I[rank * 4:(rank + 1) * 4]
    for step in range(2):
        net.store.zero_grad()
        ddp(xs, im).backward()
        assert not RT._pending_wgrad and not RT._pending_done
    nb = len(ddp.reducer.buckets)
    assert nb > 3 and ddp.reducer.launches == 2 * nb, (nb, ddp.reducer.launches)       # every bucket once per step: no second round
    assert sorted(order[:nb]) == list(range(nb)) and order[:nb] == order[nb:]
    assert late and all(0 < n for n in late)      # the parked announcement arrived after other buckets had gone, from inside Reducer.wait
    gathered = [None] * world
    dist.all_gather_object(gathered, order)
    assert all(o == gathered[0] for o in gathered), gathered
    torch.save(dict(G=net.store.G.clone(), order=order), os.path.join(tmp, f"o{rank}.pt"))
    dist.destroy_process_group()


def test_bucket_launch_order_is_identical_on_four_ranks_with_a_parked_gradient(tmp_path):
    world, port = 4, 33000 + os.getpid() % 2000
    mp.start_processes(_order_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    rs = [torch.load(tmp_path / f"o{r}.pt") for r in range(world)]
    assert all(r["order"] == rs[0]["order"] for r in rs) and all(torch.equal(r["G"], rs[0]["G"]) for r in rs)
    from crog_amd.runtime import ParamStore
    torch.manual_seed(100)
    net = Net()
    st = ParamStore(net, torch.device("cpu"))
    g = torch.Generator().manual_seed(7)
    X, I = torch.randn(16, 12, generator=g), torch.randn(16, 2, 6, 6, generator=g)
    (sum(net(X[4 * r:4 * r + 4], I[4 * r:4 * r + 4]) for r in range(4)) / 4).backward()
    assert torch.allclose(st.G, rs[0]["G"], atol=1e-6, rtol=1e-5)        # the parked gradient made it into its bucket


class _FakeLib:
    """Stands in for the C ABI (crog_comm_*) in the set-up protocol test: every call succeeds unless told to fail on this rank."""

    def __init__(self, fail_uid=False, fail_init=False, fail_handle=False, fail_connect=False, fail_init_rccl=False):
        self.fail = dict(uid=fail_uid, init=fail_init, handle=fail_handle, connect=fail_connect, init_rccl=fail_init_rccl)
        self.destroyed = 0

    def crog_last_error(self):
        return b"fake failure"

    def crog_comm_unique_id(self, _buf):
        return -2 if self.fail["uid"] else 0

    def crog_comm_init(self, rank, world, uid, comm_p):
        if self.fail["init"] or (self.fail["init_rccl"] and uid is not None):
            return -2
        comm_p._obj.value = 0x1234          # ctypes.byref(c_void_p): the handle the real library would write
        return 0

    def crog_comm_peer_handle(self, comm, slot, buf):
        return -2 if self.fail["handle"] else 0

    def crog_comm_peer_connect(self, comm, blob):
        assert len(blob) == 2 * 64
        return -2 if self.fail["connect"] else 0

    def crog_comm_destroy(self, _c):
        self.destroyed += 1
        return 0


def _setup_worker(rank, world, port, tmp):
    """The direct communicator's set-up verdict is collective (advisor, round 2): whatever fails, on whichever single rank, BOTH ranks
    return None and nobody is left waiting in a collective the other never enters."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crog_amd import _lib, rccl
    outcomes = []
    scenarios = [("library missing on rank 1", dict(load_fail=1)), ("unique id fails on rank 0", dict(uid=0)), ("comm init fails on rank 0", dict(init=0)),
                 ("comm init fails on rank 1", dict(init=1)), ("mailbox allocation fails on rank 1", dict(handle=1)),
                 ("opening a peer mailbox fails on rank 0", dict(connect=0)),
                 ("RCCL init fails on rank 1, the mailbox alone survives", dict(init_rccl=1)), ("all fine", dict())]
    for name, sc in scenarios:
        fake = _FakeLib(**{"fail_" + k: v == rank for k, v in sc.items() if k != "load_fail"})
        _lib._lib = fake                      # what K.lib() / check() return from now on
        lib = None
        if sc.get("load_fail") == rank:
            class Missing:
                def __getattr__(self, k):
                    raise OSError("libcrog_hip.so: cannot open shared object file")
            lib = Missing()
        comm, err = rccl.DirectComm.create(None, device="cpu", rccl=True, peer=True, lib=lib or fake)
        outcomes.append((name, comm is not None, repr(err)[:60]))
        if "mailbox alone" in name:      # every rank retried without RCCL (collective verdict) and got the peer-only communicator
            assert comm is not None and err is None and not comm.has_rccl and comm.has_peer, (rank, name)
        elif name != "all fine":
            assert comm is None and err is not None, (rank, name)
        else:
            assert comm is not None and err is None and comm._h.value == 0x1234 and comm.has_rccl and comm.has_peer
        dist.barrier()          # both ranks are still in step: no rank is stuck in a collective of the previous scenario
    torch.save(outcomes, os.path.join(tmp, f"setup{rank}.pt"))
    dist.destroy_process_group()


def test_direct_comm_setup_verdict_is_collective(tmp_path):
    world, port = 2, 31000 + os.getpid() % 2000
    mp.start_processes(_setup_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    o0, o1 = torch.load(tmp_path / "setup0.pt"), torch.load(tmp_path / "setup1.pt")
    assert [(n, ok) for n, ok, _ in o0] == [(n, ok) for n, ok, _ in o1] and len(o0) == 8
