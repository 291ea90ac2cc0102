cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/raw -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_sq")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.Counter()
for f in glob.glob(os.path.join(out, "raw", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            agg[k][row.get("Counter_Name", "?")] += float(row.get("Counter_Value", 0) or 0)
            if row.get("Counter_Name") == "SQ_WAVES":
                disp[k] += 1
    os.remove(f)
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"]
with open(os.path.join(out, "pmc_SQ_by_kernel.csv"), "w") as o:
    o.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        o.write("%s,%d,%s\n" % (k, disp[k], ",".join("%.0f" % v.get(n, 0) for n in names)))
PY
head -12 $OUT/pmc_SQ_by_kernel.csv
