export SWZ_MD_TIME_LIMIT=20
for cfg in "SWZ_MD_MAX_POP=160" "SWZ_MD_MAX_POP=400" "SWZ_MD_MAX_POP=3000" "SWZ_MD_COARSEN=1 SWZ_MD_COARSEN_MIN=0"; do
echo "== $cfg"
env $cfg SWZ_DEBUG=1 timeout 600 python bench.py --points 100000000 --batches 10 --sampler MIN_DISTANCE --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep -E "level -1 on keys|ms_per_step" | cut -c1-200 | tail -3
done
