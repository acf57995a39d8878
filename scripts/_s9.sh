mkdir -p gpurun_out
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"achieved": [0-9.]*'; }
export CROG_LIB=crog_amd/libcrog_hip_x.so
for rep in 1 2; do
echo "== default $(run | tr '\n' ' ')"
echo "== fat512 $(CROG_GEMM_CONV_TILE=f run | tr '\n' ' ')"
echo "== fat256 $(CROG_GEMM_CONV_TILE=f CROG_GEMM_CONV_MIN=256 run | tr '\n' ' ')"
echo "== 8w512 $(CROG_GEMM_CONV_TILE=8 run | tr '\n' ' ')"
echo "== 8w256 $(CROG_GEMM_CONV_TILE=8 CROG_GEMM_CONV_MIN=256 run | tr '\n' ' ')"
done
