# Collects hardware counters per kernel for one cold step of the default bench (separate --pmc passes, kernel trace off)
# and prints per-kernel sums: tools/pmc_probe.sh "CTR_A CTR_B" "CTR_C ..."   (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_probe
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 ${PMC_BENCH_ARGS} > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_probe")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
detail = os.environ.get("PMC_DETAIL", "")
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0][:60]
            agg[k][row.get("Counter_Name", "?")] += float(row.get("Counter_Value", 0) or 0)
            d = (k, row.get("Dispatch_Id"))
            if d not in seen:
                seen.add(d)
            if detail and detail in k:  # PMC_DETAIL=<substring>: the counters of every dispatch of those kernels
                per[(k, int(row.get("Dispatch_Id") or 0))][row.get("Counter_Name", "?")] += float(row.get("Counter_Value", 0) or 0)
    os.remove(f)
with open(os.path.join(out, "summary.txt"), "w") as o:
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        o.write(k + "\n")
        for cn, v in sorted(cs.items()):
            o.write("    %-40s %.4g\n" % (cn, v))
    for (k, d), cs in sorted(per.items(), key=lambda kv: kv[0][1]):
        o.write("dispatch %d %s: %s\n" % (d, k, "  ".join("%s=%.4g" % (cn, v) for cn, v in sorted(cs.items()))))
print(open(os.path.join(out, "summary.txt")).read()[:6000])
PY
