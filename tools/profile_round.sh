# Rounds 5-6: collects the round's evidence into gpurun_out/profile/ (tools/install_profiles.sh copies it to profiles/rNN):
# bench lines, rocprofv3 kernel summaries and HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) of
# the headline (exact), the property mode, a grid sampler and the multi-batch tiler at the reference's default operating
# point (MIN_DISTANCE, FAST, batches of 10 M points).  PARTS (environment) selects: lines stats pmc fullsize (default: all).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile
PARTS=${PARTS:-lines stats pmc fullsize}
# (nothing of an earlier run or round may survive next to this run's stamp: the lines part starts from an empty directory)
if [[ $PARTS == *lines* ]]; then rm -rf $OUT; fi
mkdir -p $OUT
sha=$(python -c "import bench; print(bench.library_source_sha16())")
if [ -f $OUT/source_sha16.txt ] && [ "$(cat $OUT/source_sha16.txt)" != "$sha" ]; then echo "profile_round: $OUT holds results of other kernel sources ($(cat $OUT/source_sha16.txt) vs $sha): run PARTS=lines first" >&2; exit 1; fi
echo $sha > $OUT/source_sha16.txt
B() { tag=$1; shift; timeout 1200 python bench.py "$@" 2>> $OUT/bench.err | grep '^{' | tail -1 > $OUT/bench_$tag.json; echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$tag.json | head -1)"; }
if [[ $PARTS == *lines* ]]; then
  : > $OUT/bench.err
  B 1B_min_distance                                   # (--md-mode both is the default: exact = value, property beside it)
  B 100M_grid_center --points 100000000 --sampler GRID_CENTER --steps 5 --warmup 2 --cpu-sample 2000000
  for s in RANDOM_GRID GRID_CENTER JITTERED; do B 1B_$s --sampler $s --steps 3 --warmup 1 --cpu-sample 0; done
  B 1B_min_distance_FAST --strategy FAST --steps 2 --warmup 1 --cpu-sample 0
  # the reference's DEFAULT operating point (executable/main.cpp:233-236, 249-251, 299-301): MIN_DISTANCE, FAST, batches of
  # 10 M points -- 1 B points in 100 batches, cut out of the whole cloud and as x-y tiles, exact and property mode
  # (bench.py keeps warming up until a data set leaves the workspace as it found it: a multi-GB hipMalloc in the middle of
  # a data set stalls it for 0.25-2 s, tools/default_op_probe.py)
  # Every multi-batch line reports `first_data_set_ms`, and the driver wipes the memory a process frees when it exits: a process
  # that starts right after one that held 100 GB waits for that inside its first allocations (tools/lists/first_data_set.txt: first
  # data set 0.95 s on a fresh device, 4.2 s right after the headline bench, 0.95 s half a minute later) -- hence the pauses.
  MB() { sleep 30; B "$@"; }
  for o in uniform tiles; do
    MB 1B_100batches_MIN_DISTANCE_FAST_$o --points 1000000000 --batches 100 --batch-order $o --strategy FAST --steps 2 --warmup 1 --cpu-sample 0 --md-mode exact
    MB 1B_100batches_MIN_DISTANCE_FAST_${o}_property --points 1000000000 --batches 100 --batch-order $o --strategy FAST --steps 2 --warmup 1 --cpu-sample 0 --md-mode property
    MB 1B_100batches_MIN_DISTANCE_$o --points 1000000000 --batches 100 --batch-order $o --steps 1 --warmup 1 --cpu-sample 0 --md-mode exact
    MB 1B_100batches_RANDOM_GRID_$o --points 1000000000 --batches 100 --batch-order $o --sampler RANDOM_GRID --steps 2 --warmup 1 --cpu-sample 0
  done
  MB 500M_5batches_staged --points 500000000 --batches 5 --staged --payload rgb,intensity --steps 2 --warmup 1 --cpu-sample 0
  timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE > $OUT/clustered_100M.txt 2>> $OUT/bench.err
  timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE property >> $OUT/clustered_100M.txt 2>> $OUT/bench.err
  timeout 600 python tools/clustered_probe.py 100000000 GRID_CENTER >> $OUT/clustered_100M.txt 2>> $OUT/bench.err
  B group_driver_8shards_1device --driver group --gpus 8 --group-devices 1 --points 25000000 --steps 2 --warmup 1
fi
if [[ $PARTS == *fullsize* ]]; then
  rm -f $GRAFT_REPO_ROOT/gpurun_out/fullsize_verification.log
  timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_min_distance_property.py::test_property_mode_full_size -q > $OUT/fullsize_pytest.txt 2>&1
  cp $GRAFT_REPO_ROOT/gpurun_out/fullsize_verification.log $OUT/fullsize_verification_1B.log 2>/dev/null
  tail -3 $OUT/fullsize_pytest.txt
fi
cd /tmp && export TMPDIR=/tmp
S() { dir=$1; shift; rm -rf $OUT/$dir; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$dir -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --also "" "$@" > $OUT/$dir.json 2>/dev/null; find $OUT/$dir -name "*kernel_trace*" -delete; f=$(find $OUT/$dir -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$dir/bench_kernel_stats.csv; }
P() { ctr=$1; dir=$2; shift 2; rm -rf $OUT/$dir; timeout 1200 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$dir -o p -- python3 $GRAFT_REPO_ROOT/bench.py --also "" "$@" > /dev/null 2>&1; }
if [[ $PARTS == *stats* ]]; then
  S stats --cpu-sample 0 --md-mode exact
  cp $OUT/stats.json $OUT/stats_run.json
  S stats_prop --cpu-sample 0 --md-mode property
  S stats_gc --sampler GRID_CENTER --cpu-sample 0
  S stats_mbmd --points 1000000000 --batches 100 --batch-order tiles --strategy FAST --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
fi
if [[ $PARTS == *pmc* ]]; then
  P FETCH_SIZE pmc_fetch --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
  P WRITE_SIZE pmc_write --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
  P FETCH_SIZE pmc_fetch_prop --steps 1 --warmup 0 --cpu-sample 0 --md-mode property
  P WRITE_SIZE pmc_write_prop --steps 1 --warmup 0 --cpu-sample 0 --md-mode property
  P FETCH_SIZE pmc_fetch_gc --sampler GRID_CENTER --steps 1 --warmup 0 --cpu-sample 0
  P WRITE_SIZE pmc_write_gc --sampler GRID_CENTER --steps 1 --warmup 0 --cpu-sample 0
  P FETCH_SIZE pmc_fetch_mbmd --points 1000000000 --batches 100 --batch-order tiles --strategy FAST --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
  P WRITE_SIZE pmc_write_mbmd --points 1000000000 --batches 100 --batch-order tiles --strategy FAST --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
  P "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" pmc_sq1_prop --steps 1 --warmup 0 --cpu-sample 0 --md-mode property
  P "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" pmc_sq2_prop --steps 1 --warmup 0 --cpu-sample 0 --md-mode property
  P "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" pmc_sq1 --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
  P "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY" pmc_sq2 --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact
  python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profile")
def per_kernel(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for d in dirs:
        for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name", "?").split("(")[0]
                    agg[k][row.get("Counter_Name", "?")] += float(row.get("Counter_Value", 0) or 0)
                    disp[k].add((d, row.get("Dispatch_Id")))
            os.remove(f)
    return agg, disp
for tag, suffix in (("", ""), ("_prop", "_property_mode"), ("_gc", "_GRID_CENTER"), ("_mbmd", "_100batches_MIN_DISTANCE_FAST")):
    for ctr in ("fetch", "write"):
        agg, disp = per_kernel(["pmc_%s%s" % (ctr, tag)])
        if not agg:
            continue
        with open(os.path.join(out, "pmc_%s_SIZE_by_kernel%s.csv" % (ctr.upper(), suffix)), "w") as o:
            o.write("kernel,dispatches,sum_counter_value_KiB\n")
            for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
                o.write("%s,%d,%.1f\n" % (k, len(disp[k]), sum(cs.values())))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES"]
for tag, suffix in (("", ""), ("_prop", "_property_mode")):
    agg, disp = per_kernel(["pmc_sq1" + tag, "pmc_sq2" + tag])
    if not agg:
        continue
    with open(os.path.join(out, "pmc_SQ_by_kernel%s.csv" % suffix), "w") as o:
        o.write("kernel,dispatches_per_pass," + ",".join(names) + "\n")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            o.write("%s,%d,%s\n" % (k, len(disp[k]) // 2, ",".join("%.0f" % v.get(n, 0) for n in names)))
PY
fi
ls $OUT | head -80; tail -3 $OUT/bench.err
