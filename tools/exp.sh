run() { echo "== $*"; env "$@" SWZ_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --cpu-sample 0 2>&1 >/dev/null | grep "keys: sweep" | tail -3 | sed 's/.*level/level/'; }
run SWZ_MD_PATIENT=0
run SWZ_MD_LAZY_FRAC=0.25
run SWZ_MD_LAZY_FRAC=0.9
run SWZ_MD_GROUPS=1
run SWZ_MD_GROUPS=3
run SWZ_MD_GROUPS=4
run SWZ_MD_LAZY=0
