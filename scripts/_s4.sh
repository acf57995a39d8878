mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/t4_kern.log
(cd .old_tree && TAG=old python scripts/bench_shapes.py) > gpurun_out/shapes4_old.log 2>&1
TAG=new python scripts/bench_shapes.py > gpurun_out/shapes4_new.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA256=1 TAG=new256 python scripts/bench_shapes.py > gpurun_out/shapes4_256.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA_TILE=x TAG=new8x python scripts/bench_shapes.py > gpurun_out/shapes4_8x.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA_TILE=f TAG=newfat python scripts/bench_shapes.py > gpurun_out/shapes4_fat.log 2>&1
python -m pytest tests -m gpu -q --deselect tests/test_kernels_gpu.py --tb=short 2>&1 | grep -v "^$" | tail -80 > gpurun_out/t4_model.log
(cd .old_tree && python bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/b4_old.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b4_new.log 2>&1
tail -n 3 gpurun_out/t4_kern.log gpurun_out/t4_model.log
