cd $GRAFT_REPO_ROOT
for ab in 4; do
  echo "ablate $ab"
  SWZ_MD_ABLATE=$ab SWZ_DEBUG=1 timeout 900 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "sweep:|rounds" | sed 's/, [0-9]* activations.*//' | cut -c1-140
done
