R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_now
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_now -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-kernel none > $R/gpurun_out/p_now.log 2>&1
cd $R
f=$(find gpurun_out/prof_now -name "*.db" | head -1)
python scripts/timeline_db.py $f 2
python scripts/prof_db.py $f 6 30
rm -f $f
