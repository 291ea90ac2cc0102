export SWZ_MD_TIME_LIMIT=20
for cfg in "SWZ_MD_DENSITY=7" "SWZ_MD_KEYS_RG=4" "SWZ_MD_KEYS_RG=8"; do
echo "== $cfg"
env $cfg SWZ_DEBUG=1 timeout 200 python bench.py --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep -E "sweep|ms_per_step|sparse" | tail -6 | cut -c1-200
done
