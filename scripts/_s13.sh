mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b13_torchrun.log 2>&1
CROG_FORCE_DDP=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b13_forced.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke13.log 2>&1
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -12 > gpurun_out/t13.log
grep -o '"ms_per_step": [0-9.]*\|"collectives_per_step": {[^}]*}' gpurun_out/b13_torchrun.log gpurun_out/b13_forced.log; tail -2 gpurun_out/smoke13.log; grep "passed\|failed" gpurun_out/t13.log
