mkdir -p gpurun_out
python -m pytest tests/test_engine_gpu.py tests/test_fulldepth_gpu.py tests/test_ddp2_gpu.py -m gpu -q --tb=short -s 2>&1 | grep -v "^$" > gpurun_out/t3.log
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA256=1 TAG=dma256 python scripts/bench_shapes.py > gpurun_out/shapes_256b.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA_TILE=x TAG=dma8x python scripts/bench_shapes.py > gpurun_out/shapes_8xb.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so TAG=x-default python scripts/bench_shapes.py > gpurun_out/shapes_xdef.log 2>&1
tail -n 5 gpurun_out/t3.log
