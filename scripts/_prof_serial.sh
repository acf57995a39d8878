R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_now_serial
CROG_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_now_serial -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --roofline-kernel none > $R/gpurun_out/p_now_serial.log 2>&1
cd $R
f=$(find gpurun_out/prof_now_serial -name "*kernel_trace.csv" | head -1)
python scripts/trace_groups.py $f 5 50
