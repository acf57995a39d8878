mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p_r02.log 2>&1
CROG_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_serial -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p_r02_serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r02_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p_r02_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r02_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p_r02_write.log 2>&1
cd $R
find gpurun_out/prof_r02 gpurun_out/prof_r02_serial gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write -type f | head -30
du -sh gpurun_out/prof_r02 gpurun_out/pmc_r02_fetch
python -m pytest tests/test_ddp2_gpu.py "tests/test_kernels_gpu.py::test_gemm_batched_heads" -m gpu -q --tb=short -s 2>&1 | grep -v "^$" | tail -12
