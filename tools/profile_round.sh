# Collects this round's evidence into gpurun_out/profile/: the bench line, the rocprofv3 kernel summary of
# the same command, and HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile
rm -rf $OUT; mkdir -p $OUT
python -c "import bench; print(bench.library_source_sha16())" > $OUT/source_sha16.txt
timeout 600 python bench.py > $OUT/bench_1B_min_distance.json 2> $OUT/bench.err   # (--md-mode both is the default: exact = value, property beside it)
timeout 600 python bench.py --points 100000000 --sampler GRID_CENTER --steps 5 --warmup 2 --cpu-sample 2000000 > $OUT/bench_100M_grid_center.json 2>> $OUT/bench.err
for s in RANDOM_GRID GRID_CENTER JITTERED; do
  timeout 600 python bench.py --sampler $s --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_1B_$s.json 2>> $OUT/bench.err
done
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --payload rgb,intensity > $OUT/bench_1B_min_distance_payload.json 2>> $OUT/bench.err
# BASELINE config 5, single-GPU shape: batches from pinned host memory, attributes along, copies under the kernels
timeout 600 python bench.py --points 500000000 --batches 5 --staged --payload rgb,intensity --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_500M_5batches_staged.json 2>> $OUT/bench.err
timeout 600 python bench.py --points 500000000 --batches 5 --staged --payload rgb,intensity --sampler RANDOM_GRID --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_500M_5batches_staged_RANDOM_GRID.json 2>> $OUT/bench.err
timeout 600 python bench.py --strategy FAST --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_1B_min_distance_FAST.json 2>> $OUT/bench.err
SWZ_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_1B_min_distance_sharded_driver_1rank.json 2>> $OUT/bench.err
timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE > $OUT/clustered_100M.txt 2>> $OUT/bench.err
timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE property >> $OUT/clustered_100M.txt 2>> $OUT/bench.err
timeout 600 python tools/clustered_probe.py 100000000 GRID_CENTER >> $OUT/clustered_100M.txt 2>> $OUT/bench.err
# the reference's default batch size (10 M points): 100 M points in 10 batches, through the multi-batch tiler
for s in MIN_DISTANCE RANDOM_GRID; do
  timeout 900 python bench.py --points 100000000 --batches 10 --sampler $s --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_100M_10batches_$s.json 2>> $OUT/bench.err
done
# ... and the whole 1 B points in 100 batches of 10 M: cut out of the whole cloud (every batch reaches every node) and as
# x-y tiles (spatially coherent, the way LAS tiles arrive); one warm-up data set first (the cold run is mostly hipMalloc)
for s in MIN_DISTANCE RANDOM_GRID; do
  for o in uniform tiles; do
    timeout 1200 python bench.py --points 1000000000 --batches 100 --batch-order $o --sampler $s --steps 1 --warmup 1 --cpu-sample 0 --md-mode exact > $OUT/bench_1B_100batches_${s}_$o.json 2>> $OUT/bench.err
  done
done
# the one-process C++ driver through bench.py (all shards on this one device): single batch, and FAST in three batches
timeout 600 python bench.py --driver group --gpus 8 --group-devices 1 --points 25000000 --steps 2 --warmup 1 > $OUT/bench_group_driver_8shards_1device.json 2>> $OUT/bench.err
timeout 600 python bench.py --driver group --gpus 8 --group-devices 1 --points 25000000 --batches 3 --strategy FAST --steps 2 --warmup 1 > $OUT/bench_group_driver_8shards_1device_FAST_3batches.json 2>> $OUT/bench.err
# ... and ACCURATE in three batches: every batch's MIN_DISTANCE root swept by all shards at once (new in round 4) / in turns
timeout 600 python bench.py --driver group --gpus 8 --group-devices 1 --points 25000000 --batches 3 --steps 2 --warmup 1 > $OUT/bench_group_driver_8shards_1device_3batches_joint.json 2>> $OUT/bench.err
SWZ_GROUP_JOINT_ROOT=0 timeout 600 python bench.py --driver group --gpus 8 --group-devices 1 --points 25000000 --batches 3 --steps 2 --warmup 1 > $OUT/bench_group_driver_8shards_1device_3batches_turns.json 2>> $OUT/bench.err
# a batch sharded over 8 and 2 contexts of this one GPU from one C++ process: MIN_DISTANCE root swept by all shards at once / in turns
bash tools/group_bench.sh > $OUT/group_joint_root_vs_turns.txt 2>> $OUT/bench.err
# full-size verification at the size the GPU has room for (1 B points on a 288 GB part); the test logs what it verified
rm -f $GRAFT_REPO_ROOT/gpurun_out/fullsize_verification.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q > $OUT/fullsize_pytest.txt 2>&1
cp $GRAFT_REPO_ROOT/gpurun_out/fullsize_verification.log $OUT/fullsize_verification_1B.log 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --md-mode exact > $OUT/stats_run.json 2>/dev/null
find $OUT/stats -name "*kernel_trace*" -delete
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact > /dev/null 2>&1
# the same three passes for a grid sampler, the property mode and the multi-batch tiler (kernel summaries only for the last two)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gc -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --sampler GRID_CENTER --cpu-sample 0 > /dev/null 2>&1
find $OUT/stats_gc -name "*kernel_trace*" -delete
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_gc -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --sampler GRID_CENTER --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_gc -o write -- python3 $GRAFT_REPO_ROOT/bench.py --sampler GRID_CENTER --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_prop -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --md-mode property > /dev/null 2>&1
find $OUT/stats_prop -name "*kernel_trace*" -delete
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mb -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --points 1000000000 --batches 100 --sampler RANDOM_GRID --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
find $OUT/stats_mb -name "*kernel_trace*" -delete
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profile")
for tag in ("fetch", "write", "fetch_gc", "write_gc"):
    for f in glob.glob(os.path.join(out, "pmc_%s" % tag, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?").split("(")[0]
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0) or 0)
        with open(os.path.join(out, "pmc_%s_SIZE_by_kernel%s.csv" % (tag.split("_")[0].upper(), "_GRID_CENTER" if tag.endswith("_gc") else "")), "w") as o:
            o.write("kernel,dispatches,sum_counter_value_KiB\n")
            for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                o.write("%s,%d,%.1f\n" % (k, n, v))
        os.remove(f)
PY
ls -la $OUT; head -c 600 $OUT/bench_1B_min_distance.json; echo; cat $OUT/bench.err | tail -3
