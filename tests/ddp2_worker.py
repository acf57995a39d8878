"""Child process of tests/test_ddp2_gpu.py: one rank of a world-size-`world` data-parallel run of the tiny CROG on cuda:0
(every rank shares the one GPU of the test box; the collectives go through gloo on CUDA tensors), or the single-process run
on the whole batch when world == 1.  Mirrors train_crog.py:113-121,154-156 + crog_engine.py:72-84: SyncBatchNorm conversion,
DistributedDataParallel(find_unused_parameters=True), Adam, forward -> zero_grad -> backward -> step, twice."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _compact(key, t):
    """A 66.6 M-float buffer as its SHA-1 (bit-equality of the whole buffer across ranks / runs) + every 16th element (norms, cosines): 17 MB
    instead of 266 MB per buffer and rank on disk."""
    import hashlib
    a = t.detach().cpu().numpy()
    return {key: np.ascontiguousarray(a[::16]), key + "_sha": np.frombuffer(hashlib.sha1(a.tobytes()).digest(), dtype=np.uint8).copy()}


def _install_trace(path):
    """Debugging aid (scripts/many_rank_probe.py, CROG_WORKER_TRACE=1): one line per collective this rank issues - time, sequence number,
    kind (stat = a BatchNorm statistics exchange launched from Python, fuse = one handed to its producing kernel, bucket = a gradient
    bucket's all-reduce with its index, wait = Reducer.wait) - flushed as written, so that the ranks' sequences can be diffed and a
    rank that stops can be told from one that is slow."""
    import time
    from crog_amd import parallel as PP
    f, t0, seq = open(path, "w"), time.time(), [0]

    def log(kind, detail=""):
        seq[0] += 1
        f.write(f"{time.time() - t0:9.3f} {seq[0]:5d} {kind} {detail}\n")
        f.flush()

    def wrap(cls, name, kind, detail):
        orig = getattr(cls, name)

        def w(self, *a, **k):
            log(kind + "<", detail(self, *a, **k))
            r = orig(self, *a, **k)
            log(kind + ">", "" if r is None or kind != "fuse" else "in-kernel")
            return r
        setattr(cls, name, w)
    wrap(PP.SyncBNComm, "all_reduce_sum", "stat", lambda self, t: f"n={t.numel()}")
    wrap(PP.SyncBNComm, "fuse_ptr", "fuse", lambda self, n: f"n={n}")
    wrap(PP.Reducer, "_launch", "bucket", lambda self, b: f"i={self.buckets.index(b)} launched={b['launched']} pending={b['pending']}")
    wrap(PP.Reducer, "wait", "wait", lambda self: f"done={self._done} unlaunched={sum(not b['launched'] for b in self.buckets)}")
    return log


def main():
    if os.environ.get("CROG_WORKER_DUMP_AFTER"):      # debugging aid (scripts/many_rank_probe.py): where is a rank that does not finish?
        import faulthandler
        every = os.environ.get("CROG_WORKER_DUMP_EVERY")
        if every:      # a dump every N seconds (a rank that moves between dumps is slow, not stuck), the process ends with the last one
            faulthandler.dump_traceback_later(int(every), repeat=True)
            import threading
            threading.Timer(int(os.environ["CROG_WORKER_DUMP_AFTER"]), lambda: os._exit(1)).start()
        else:
            faulthandler.dump_traceback_later(int(os.environ["CROG_WORKER_DUMP_AFTER"]), exit=True)
    rank, world, port, out_dir, dtype_name, gain = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], float(sys.argv[6])
    size, B, tag = int(sys.argv[7]), int(sys.argv[8]), sys.argv[9]
    from crog_amd.model import build_crog
    from crog_amd.optim import FusedAdam
    from crog_amd.parallel import DistributedDataParallel, convert_sync_batchnorm
    from crog_amd.runtime import RT
    from crog_amd.testing import seeded_state, synthetic_batch, tiny_cfg
    torch.cuda.set_device(0)
    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "tiny_crog.json")))
    cfg = tiny_cfg(input_size=size)
    model, groups = build_crog(cfg)
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    # ranks > 0 start from DIFFERENT weights: DistributedDataParallel's constructor must hand them rank 0's (train_crog.py:154)
    model.load_state_dict(seeded_state(shapes, seed=meta["seed"] + 100 * rank, residual_gain=gain))
    model = model.cuda()
    model.compute_dtype = dtype
    model.prepare()
    if os.environ.get("CROG_WORKER_SINGLE_STREAM") == "1":      # (probe: no weight-gradient / text-tower side streams)
        RT.overlap_wgrad = False
        model.overlap_text = False
    net = model
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
        dist.init_process_group("gloo", rank=rank, world_size=world)
        convert_sync_batchnorm(model)
        assert RT.comm is not None and RT.comm.world_size == world
        if os.environ.get("CROG_SYNCBN_DIRECT") == "peer":      # the statistics really travel through the hipIpc mailboxes
            assert RT.comm.direct is not None and RT.comm.direct.has_peer and RT.comm.kind == "crog_comm:peer", RT.comm.kind
        # 0.25 MiB buckets: ~1000 gradient collectives in flight per step of the tiny model (66.6 M parameters), which is what the two-rank tests want
        # to exercise; the many-rank tests pass a larger cap (CROG_WORKER_BUCKET_MB) - see tests/test_ddp2_gpu.py on what gloo costs per collective
        net = DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True, bucket_cap_mb=float(os.environ.get("CROG_WORKER_BUCKET_MB", "0.25")))
    opt = FusedAdam(groups, lr=1e-4, store=model.store)
    full = synthetic_batch(B, cfg.input_size, cfg.word_len, cfg.clip_arch["vocab_size"], seed=1234 + meta["seed"])
    per = B // world
    b = {k: v[rank * per:(rank + 1) * per].cuda() for k, v in full.items()}
    net.train()
    res = {}
    log = _install_trace(os.path.join(out_dir, f"{tag}_trace{rank}.txt")) if os.environ.get("CROG_WORKER_TRACE") == "1" else (lambda *a: None)
    names = meta["param_names"]
    for step in range(2):
        RT.manual_seed(11)
        log("step", str(step))
        preds, tgts, loss, _ = net(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
        opt.zero_grad()
        log("backward<")
        loss.backward()
        log("backward>")
        torch.cuda.synchronize()
        log("synchronized")
        if step == 0:
            params = dict(model.named_parameters())
            res["preds"] = torch.cat([p.float() for p in preds], 1).cpu().numpy()
            res["loss"] = np.float64(float(loss.detach()))
            res["grad_norms"] = np.array([float(params[n].grad.norm()) for n in names])
            res.update(_compact("G", model.store.G))
            sd = model.state_dict()
            res["bn_checksum"] = np.array([float(sd[k].double().sum()) for k in meta["bn_keys"]])
            if world > 1:
                res["n_buckets"] = np.int64(len(net.reducer.buckets))
        opt.step()
    torch.cuda.synchronize()
    res.update(_compact("P", model.store.P))
    sd = model.state_dict()
    res["bn_final"] = np.concatenate([sd[k].float().cpu().numpy().ravel() for k in meta["bn_keys"]])
    if world > 1:
        res["bucket_launches"] = np.int64(net.reducer.launches)      # over both steps: every bucket exactly once per step
        res["syncbn_launches"] = np.int64(RT.comm.calls)
        res["syncbn_in_kernel"] = np.int64(getattr(RT.comm, "fused", 0))
    if world > 1 and os.environ.get("CROG_WORKER_DUMP_AFTER"):
        print(f"rank {rank}: exchanges from Python {RT.comm.calls}, in kernels {getattr(RT.comm, 'fused', 0)}, timed out "
              f"{RT.comm.direct.timed_out() if RT.comm.direct is not None else None}, loss {res['loss']}", flush=True)
    if world > 1 and RT.comm.direct is not None:
        assert RT.comm.direct.timed_out() == 0
    np.savez(os.path.join(out_dir, f"{tag}_rank{rank}_of{world}.npz"), **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
