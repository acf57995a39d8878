R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02 $R/gpurun_out/prof_r02_serial $R/gpurun_out/pmc_r02_fetch $R/gpurun_out/pmc_r02_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p_r02.log 2>&1
CROG_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_serial -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p_r02_serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r02_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p_r02_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r02_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p_r02_write.log 2>&1
cd $R
PROFILE_NOTE="Overlapped run (main + text + weight-gradient streams): per-kernel durations include the time a kernel shares the chip with the other streams." python scripts/summarize_profile.py r02 gpurun_out/prof_r02 gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write 5 3 | tail -3
PROFILE_ENV="CROG_SINGLE_STREAM=1 " PROFILE_NOTE="Single-stream run (no weight-gradient / text-tower side streams): per-kernel durations are not inflated by concurrent kernels, so the TFLOP/s and GB/s that follow from them are the kernels' own rates." python scripts/summarize_profile.py r02_serial gpurun_out/prof_r02_serial gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write 5 3 | tail -3
mkdir -p gpurun_out/profiles_out; cp profiles/r02_summary.md profiles/r02_serial_summary.md profiles/r02_kernel_stats.csv profiles/r02_serial_kernel_stats.csv profiles/pmc_traffic.json gpurun_out/profiles_out/
f=$(find gpurun_out/prof_r02_serial -name "*kernel_trace.csv" | head -1); python scripts/trace_groups.py $f 5 40 > gpurun_out/profiles_out/serial_groups.txt
f=$(find gpurun_out/prof_r02 -name "*kernel_trace.csv" | head -1); python scripts/trace_groups.py $f 5 40 > gpurun_out/profiles_out/overlap_groups.txt
rm -rf gpurun_out/prof_r02/*/*kernel_trace.csv gpurun_out/prof_r02_serial/*/*kernel_trace.csv gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write
tail -3 gpurun_out/p_r02.log | cut -c1-300
