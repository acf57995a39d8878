mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_ddp_gpu.py -m gpu -q --tb=short -s 2>&1 | grep -v "^$" | tail -8
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|unavailable[^"]*'; }
for rep in 1 2; do
echo "== plain          $(run)"
echo "== forced direct  $(CROG_FORCE_DDP=1 run)"
echo "== forced torch   $(CROG_FORCE_DDP=1 CROG_SYNCBN_DIRECT=0 run)"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
