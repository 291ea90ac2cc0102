# BASELINE configs 4 and 5 as dry runs on ONE GPU (8 ranks / shards, scaled-down totals) and the reference's default operating
# point (FAST + MIN_DISTANCE + batches of 10 M points): gpurun_out/r5cfg/*.json
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5cfg; mkdir -p $O
run() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/$tag.json | head -1) $(grep -o '"value": [0-9.]*' $O/$tag.json | head -1)"; tail -2 $O/$tag.err | cut -c1-300; }
run config4_dryrun_8ranks_1gpu_200M --config 4 --gpus 8 --one-device --total-points 200000000 --steps 2 --warmup 1
run config5_dryrun_8ranks_1gpu_200M --config 5 --gpus 8 --one-device --total-points 200000000 --steps 1 --warmup 1
run config4_dryrun_group_8shards_1gpu_200M --config 4 --gpus 8 --driver group --group-devices 1 --total-points 200000000 --steps 2 --warmup 1
run config5_dryrun_group_8shards_1gpu_200M --config 5 --gpus 8 --driver group --group-devices 1 --total-points 200000000 --steps 2 --warmup 1
