"""Where does a many-rank one-GPU run of tests/ddp2_worker.py stop?  usage (GPU box): python scripts/many_rank_probe.py WORLD TAG [VAR=val ...]
Starts WORLD workers (fp32 tiny CROG, one sample each, SyncBatchNorm over the hipIpc mailboxes), per-rank logs in gpurun_out/mr/TAG_rankR.log;
a rank that has not finished after 70 s dumps its Python stacks and exits (CROG_WORKER_DUMP_AFTER); mailbox waits are bounded at 15 s.
Round 6: every rank also writes TAG_traceR.txt (CROG_WORKER_TRACE: one time-stamped line per collective it issues); PROBE_LIMIT=seconds
bounds the run, CROG_WORKER_DUMP_EVERY=N repeats the stack dump every N seconds."""
import os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world, tag, extra = int(sys.argv[1]), sys.argv[2], dict(a.split("=", 1) for a in sys.argv[3:])
out = os.path.join(ROOT, "gpurun_out", "mr"); os.makedirs(out, exist_ok=True)
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1])
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CROG_SYNCBN_DIRECT="peer", CROG_COMM_TIMEOUT_S="15", CROG_WORKER_DUMP_AFTER="70", CROG_WORKER_TRACE="1")
env.update(extra)
t0 = time.time()
procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp2_worker.py"), str(r), str(world), port, out, "f32", "1.0", "96", str(world), tag],
                          env=env, stdout=open(os.path.join(out, f"{tag}_rank{r}.log"), "w"), stderr=subprocess.STDOUT) for r in range(world)]
rcs = []
for p in procs:
    try:
        rcs.append(p.wait(timeout=max(1, float(os.environ.get('PROBE_LIMIT', '100')) - (time.time() - t0))))
    except subprocess.TimeoutExpired:
        p.kill(); rcs.append("killed")
import glob
for f in glob.glob(os.path.join(out, f"{tag}_rank*_of{world}.npz")):      # (66.6 M-float gradient / parameter dumps: gpurun copies back 64 MiB at most)
    os.remove(f)
print(f"{tag} {extra}: return codes {rcs} in {time.time() - t0:.0f} s", flush=True)
