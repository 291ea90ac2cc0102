cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/exp14.txt
for k in 1 2; do
echo "COARSEN=$k" >> gpurun_out/exp14.txt
SWZ_MD_COARSEN=$k SWZ_DEBUG=1 timeout 300 python bench.py --points 1000000000 --sampler MIN_DISTANCE --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "sweep:|cells|metric" | cut -c1-200 >> gpurun_out/exp14.txt
done
