run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline --roofline-kernel none 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2 3; do
echo "default         $(run)"
echo "main high       $(CROG_MAIN_PRIORITY=-1 run)"
echo "main side stream prio 0 $(CROG_MAIN_PRIORITY=0 run)"
done
