mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^$" | tail -15 > gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_full.log 2>&1
python bench.py > gpurun_out/b_full.json 2>gpurun_out/b_full.err
tail -n 4 gpurun_out/t_full.log; tail -2 gpurun_out/smoke_full.log; cat gpurun_out/b_full.json
