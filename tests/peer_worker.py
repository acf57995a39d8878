"""Child process of tests/test_ddp2_gpu.py::test_peer_mailbox_allreduce_*: one of `world` ranks sharing cuda:0.  Builds the C-ABI
communicator without RCCL (RCCL refuses two ranks on one device), runs a series of crog_syncbn_stats exchanges through the hipIpc
mailboxes and checks every one against the same all-reduce done by gloo; then times a burst."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crog_amd.rccl import SLOT_FLOATS, DirectComm
    comm, err = DirectComm.create(None, rccl=False, peer=True)
    res = dict(rank=rank, created=comm is not None, err=repr(err))
    if comm is not None:
        g = torch.Generator().manual_seed(100 + rank)
        bad = 0
        sizes = [2, 64, 128, 512, 1000, 4096, SLOT_FLOATS, 6, 2 * 2048, 130]
        for it in range(60):
            n = sizes[it % len(sizes)]
            mine0 = torch.randn(n, generator=g)
            x = mine0.cuda()
            want = x.clone()
            dist.all_reduce(want)                      # gloo on a CUDA tensor: the reference exchange
            comm.all_reduce_sum(x)
            torch.cuda.synchronize()
            if comm.timed_out():
                res["timed_out_at"] = it
                break
            # two ranks: a + b in rank order on both sides == gloo's a + b bit for bit; more ranks: gloo's tree adds in another order, so the
            # bit-exact reference is the rank-order sum of the gathered contributions (what the mailbox kernel promises: slots added 0, 1, 2, ...)
            if world == 2:
                bad += int(not torch.equal(x, want))
            else:
                parts = [torch.empty_like(mine0) for _ in range(world)]
                dist.all_gather(parts, mine0)
                acc = torch.zeros_like(mine0)
                for q in parts:
                    acc = acc + q
                bad += int(not torch.equal(x.cpu(), acc)) + int(not torch.allclose(x, want, rtol=1e-5, atol=1e-5))
        res["mismatches"] = bad
        # every rank must hold the SAME bits (the sum is formed in rank order everywhere)
        y = torch.randn(4096, generator=torch.Generator().manual_seed(7 + rank)).cuda()
        comm.all_reduce_sum(y)
        torch.cuda.synchronize()
        mine = y.cpu()
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        res["identical_across_ranks"] = all(torch.equal(both[0], b) for b in both)
        # latency of a dependent chain of exchanges (what BatchNorm layers see)
        z = torch.ones(512, device="cuda")
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            comm.all_reduce_sum(z)
            z.mul_(1.0 / world)
        torch.cuda.synchronize()
        res["us_per_exchange"] = (time.perf_counter() - t0) / 200 * 1e6
        res["chain_value_ok"] = bool(torch.allclose(z, torch.ones_like(z)))
        res["timed_out"] = comm.timed_out()
        comm.close()
    json.dump(res, open(os.path.join(out, f"peer_rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
