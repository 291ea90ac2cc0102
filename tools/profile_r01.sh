# Collects the round's bench line, the rocprofv3 kernel summary of the same command and the HBM
# traffic counters (separate --pmc passes) into gpurun_out/profile_r01/.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_r01
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/stats_run.json 2>/dev/null
find $OUT/stats -name "*kernel_trace*" -delete
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
cd $OUT && ls -laR | head -40
python3 - <<'PY'
import csv, glob, collections, os
for tag in ("fetch", "write"):
    files = glob.glob(os.path.join(os.environ.get("OUT", "."), "pmc_%s" % tag, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        agg = collections.defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?").split("(")[0]
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0) or 0)
        with open(os.path.join(os.path.dirname(f), "summary_%s.csv" % tag), "w") as out:
            out.write("kernel,dispatches,sum_counter_value\n")
            for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                out.write("%s,%d,%.1f\n" % (k, n, v))
        os.remove(f)
PY
find $OUT -name "summary_*.csv" | xargs -n1 head -12
