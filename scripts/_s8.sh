mkdir -p gpurun_out
python -m pytest tests/test_preprocess_gpu.py tests/test_ssg_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -30 > gpurun_out/t8.log
python scripts/bench_preprocess.py > gpurun_out/prep.log 2>&1
tail -n 5 gpurun_out/t8.log; tail -n 2 gpurun_out/prep.log
