export SWZ_MD_TIME_LIMIT=20
timeout 600 python -m pytest tests/test_min_distance_keys.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -5
for cfg in 2048 512 128 8192; do
echo "== dense wait $cfg"
SWZ_MD_DENSE_WAIT=$cfg SWZ_DEBUG=1 timeout 300 python tools/clustered_probe.py 100000000 MIN_DISTANCE 2>&1 | grep -E "sweep|N=" | cut -c1-100 | tail -6
done
SWZ_DEBUG=1 timeout 200 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "sweep|ms_per_step" | tail -4 | cut -c1-250
