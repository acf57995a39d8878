mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd $R/.old_tree && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof5_old -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p5_old.log 2>&1
cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof5_new -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p5_new.log 2>&1
find gpurun_out/prof5_old gpurun_out/prof5_new -name "*.csv" | head
python -m pytest tests/test_engine_gpu.py tests/test_fulldepth_gpu.py "tests/test_model_gpu.py::test_checkpoint_wire_format_round_trips_with_torch_adam" -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -30 > gpurun_out/t5.log
tail -3 gpurun_out/t5.log
