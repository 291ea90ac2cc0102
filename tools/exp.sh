run() { echo "== $*"; env "$@" SWZ_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --cpu-sample 0 2>&1 >/dev/null | grep "keys: sweep" | tail -3 | sed 's/.*level/level/'; }
SWZ_MD_CHAIN=8 timeout 600 python -m pytest tests/test_min_distance_keys.py -x -q 2>&1 | tail -3
run SWZ_MD_CHAIN=2
run SWZ_MD_CHAIN=3
run SWZ_MD_CHAIN=4
