python - <<'PY'
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for p in (-1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "->", s.priority)
    except Exception as e: print(p, "error", e)
PY
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])"; }
for rep in 1 2; do
echo "prio 0   $(run)"
echo "prio 1   $(CROG_SIDE_PRIORITY=1 run)"
echo "prio -1  $(CROG_SIDE_PRIORITY=-1 run)"
done
