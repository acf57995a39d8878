mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -25 > gpurun_out/t10.log
python bench.py --steps 20 --warmup 5 > gpurun_out/b10.json 2>gpurun_out/b10.err
tail -n 4 gpurun_out/t10.log; cat gpurun_out/b10.json
