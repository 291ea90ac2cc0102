cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "SWZ_MD_LAST_HIT=1" "SWZ_MD_LAST_HIT=1 SWZ_MD_PATIENT=0"; do
echo "CFG $cfg"
env $cfg SWZ_DEBUG=1 timeout 300 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "sweep:|rounds" | grep -v sparse | tail -7 | cut -c1-150 | tee -a gpurun_out/exp13.txt
done
