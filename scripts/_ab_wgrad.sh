run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
echo "base            $(run)"
echo "t64 target 512  $(CROG_WGRAD_TARGET=512 run)"
echo "t64 target 2048 $(CROG_WGRAD_TARGET=2048 run)"
echo "t128 tgt 256    $(CROG_WGRAD_TILE=128 CROG_WGRAD_TARGET128=256 run)"
echo "t128 tgt 512    $(CROG_WGRAD_TILE=128 CROG_WGRAD_TARGET128=512 run)"
echo "t128 tgt 768    $(CROG_WGRAD_TILE=128 run)"
echo "t128 256 2strm  $(CROG_WGRAD_TILE=128 CROG_WGRAD_TARGET128=256 CROG_WGRAD_STREAMS=2 run)"
echo "conv tgt 384    $(CROG_WGRAD_TARGET_CONV=384 run)"
echo "conv tgt 1536   $(CROG_WGRAD_TARGET_CONV=1536 run)"
done
