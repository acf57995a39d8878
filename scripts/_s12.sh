mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "layernorm or batched_heads or production_conv" --tb=short 2>&1 | grep -v "^$" | tail -6
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3; do
echo "== ln-atomic $(run)"
echo "== ln-slab   $(CROG_LN_BWD_ATOMIC=0 run)"
done
python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -5
