mkdir -p gpurun_out
python -m pytest tests/test_ssg_gpu.py tests/test_engine_gpu.py tests/test_ddp2_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -40 > gpurun_out/t7.log
python scripts/bench_ssg_loss.py 64 > gpurun_out/ssgloss.log 2>&1
python scripts/bench_ssg.py > gpurun_out/ssg_trunk.log 2>&1
tail -4 gpurun_out/t7.log; tail -2 gpurun_out/ssgloss.log gpurun_out/ssg_trunk.log
