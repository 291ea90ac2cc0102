export SWZ_MD_TIME_LIMIT=20
for i in 1 2; do
SWZ_DEBUG=1 timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "sweep|ms_per_step" | tail -4 | cut -c1-250
done
