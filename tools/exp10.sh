cd $GRAFT_REPO_ROOT
for d in 2 32; do
  echo "density $d"
  SWZ_MD_DENSITY=$d SWZ_DEBUG=1 timeout 900 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "sweep:|cell_levels" | cut -c1-150
done
