mkdir -p gpurun_out
python -m pytest tests -m gpu -q --deselect tests/test_kernels_gpu.py 2>&1 | tail -120 > gpurun_out/t2_model.log
(cd .old_tree && python scripts/loss_trace.py 14) > gpurun_out/trace_old.log 2>&1
python scripts/loss_trace.py 14 > gpurun_out/trace_new.log 2>&1
python scripts/loss_trace.py 14 > gpurun_out/trace_new2.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_np.so python scripts/ablate_gemm.py > gpurun_out/ablate.log 2>&1
python scripts/bench_shapes.py > gpurun_out/shapes_base.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA_TILE=f TAG=fat python scripts/bench_shapes.py > gpurun_out/shapes_fat.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA256=1 TAG=dma256 python scripts/bench_shapes.py > gpurun_out/shapes_256.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA_TILE=x TAG=dma8x python scripts/bench_shapes.py > gpurun_out/shapes_8x.log 2>&1
tail -n 3 gpurun_out/t2_model.log gpurun_out/trace_old.log gpurun_out/trace_new.log
