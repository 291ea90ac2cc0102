export SWZ_MD_TIME_LIMIT=20
for cfg in "SWZ_MD_BIG=0 SWZ_MD_GROUPS=1" "SWZ_MD_GROUPS=1"; do
echo "== $cfg"
env $cfg SWZ_DEBUG=1 timeout 300 python tools/clustered_probe.py 100000000 MIN_DISTANCE 2>&1 | grep -E "sweep|N=" | cut -c1-100 | tail -6
done
SWZ_DEBUG=1 timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "sweep|ms_per_step" | tail -4 | cut -c1-200
