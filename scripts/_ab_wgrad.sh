run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline --roofline-kernel none 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
echo "new defaults    $(run)"
echo "old defaults    $(CROG_WGRAD_TILE=64 CROG_WGRAD_TARGET128=768 CROG_WGRAD_TARGET_CONV=768 run)"
echo "conv 384        $(CROG_WGRAD_TARGET_CONV=384 run)"
echo "conv 640        $(CROG_WGRAD_TARGET_CONV=640 run)"
echo "t128 192        $(CROG_WGRAD_TARGET128=192 run)"
echo "t128 320        $(CROG_WGRAD_TARGET128=320 run)"
echo "2 streams       $(CROG_WGRAD_STREAMS=2 run)"
done
