mkdir -p gpurun_out
for t in old new; do
  d=$PWD; [ $t = old ] && d=$PWD/.old_tree
  (cd $d && python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 16 | grep -o '"ms_per_step": [0-9.]*') > gpurun_out/b6_${t}_b16.log 2>&1
  (cd $d && python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 8 | grep -o '"ms_per_step": [0-9.]*') > gpurun_out/b6_${t}_b8.log 2>&1
  (cd $d && CROG_OVERLAP_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*') > gpurun_out/b6_${t}_nowg.log 2>&1
  (cd $d && python bench.py --steps 10 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*') > gpurun_out/b6_${t}_b32.log 2>&1
done
CROG_LIB=crog_amd/libcrog_hip_x.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/b6_x_def.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA_TILE=f python bench.py --steps 10 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/b6_x_fat.log 2>&1
CROG_LIB=crog_amd/libcrog_hip_x.so CROG_GEMM_DMA256=a python bench.py --steps 10 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' > gpurun_out/b6_x_256a.log 2>&1
grep . gpurun_out/b6_*.log
