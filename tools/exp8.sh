cd $GRAFT_REPO_ROOT/schwarzwald_amd/csrc
make clean >/dev/null; make -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DSWZ_MD_STATS" 2>&1 | grep -E "error" 
cd $GRAFT_REPO_ROOT
SWZ_DEBUG=1 timeout 900 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "swz\]" 
