L=crog_amd/variants/libcrog_nopk_bperm.so
echo "=== new default build: probe"; python scripts/pk_probe.py 2000 wgrad 2>&1 | tail -1
echo "=== new default build, all streams B=8 N=24"; DET_VARIANT=all python scripts/det_stress.py 8 0.1 24 2>&1 | grep -E "runs differ|^run" | cut -c1-300
echo "=== ds_bpermute reductions WITHOUT packed fp32, all streams B=8 N=24"; CROG_LIB=$L DET_VARIANT=all python scripts/det_stress.py 8 0.1 24 2>&1 | grep -E "runs differ|^run" | cut -c1-300
echo "=== ds_bpermute reductions WITHOUT packed fp32, all streams B=2 N=24"; CROG_LIB=$L DET_VARIANT=all python scripts/det_stress.py 2 0.1 24 2>&1 | grep -E "runs differ|^run" | cut -c1-300
echo "=== det_probe (B=4, p=0, N=40, the configuration that showed 26 of 238) with the bperm lib"; CROG_LIB=$L python scripts/det_probe.py 4 0.0 40 2>&1 | tail -4 | cut -c1-300
