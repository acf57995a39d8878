"""How fast are gloo collectives on CUDA tensors when WORLD processes share ONE GPU (the rig of tests/test_ddp2_gpu.py)?
usage (GPU box): python scripts/gloo_cuda_rate.py WORLD [busy]   -> per-collective milliseconds, blocking small ones (the BatchNorm
statistics' shape: 2 x 64 ... 2 x 512 floats) and asynchronous 256 KiB ones waited for in a batch (the gradient buckets' shape); "busy" keeps
a chain of small kernels queued on the current stream between the collectives, as a backward pass does."""
import os, socket, subprocess, sys, time
if len(sys.argv) >= 2 and sys.argv[1] == "worker":
    rank, world, port, busy = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "busy"
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g2 = dist.new_group()
    small = [torch.ones(2 * c, device="cuda") for c in (64, 128, 256, 512)]
    big = [torch.ones(1 << 16, device="cuda") for _ in range(16)]
    work = torch.ones(1 << 20, device="cuda")
    side = torch.cuda.Stream()
    for rep in range(2):
        torch.cuda.synchronize(); dist.barrier(); t0 = time.time()
        n = 100
        for i in range(n):
            if busy:
                for _ in range(4):
                    work.mul_(1.0001)
            dist.all_reduce(small[i % 4], group=g2)
        torch.cuda.synchronize(); t1 = time.time()
        works = []
        for i in range(64):
            if busy:
                for _ in range(4):
                    work.mul_(1.0001)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                works.append(dist.all_reduce(big[i % 16], async_op=True))
            if i % 4 == 3:
                dist.all_reduce(small[i % 4], group=g2)       # statistics exchanges interleaved with buckets in flight
        t2 = time.time()
        for w in works:
            w.wait()
        torch.cuda.synchronize(); t3 = time.time()
        if rank == 0:
            print(f"world {world} {'busy' if busy else 'idle'} pass {rep}: blocking small all-reduce {(t1 - t0) / n * 1e3:.2f} ms each; 64 async 256 KiB buckets + 16 small: "
                  f"issue {(t2 - t1) * 1e3:.1f} ms, drained after {(t3 - t1) * 1e3:.1f} ms", flush=True)
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
world, busy = int(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "idle")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1])
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(world), port, busy], env=env) for r in range(world)]
t0 = time.time()
for p in procs:
    try:
        p.wait(timeout=max(1, 150 - (time.time() - t0)))
    except subprocess.TimeoutExpired:
        p.kill(); print("killed a rank after 150 s", flush=True)
