export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_ddp_gpu.py tests/test_ddp2_gpu.py -m gpu -q --tb=short 2>&1 | grep "passed\|failed\|error" | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
CROG_FORCE_DDP=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"collectives_per_step": {[^}]*}'
CROG_FORCE_DDP=1 CROG_GRAD_PAYLOAD=bf16 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*'
