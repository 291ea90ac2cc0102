cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --points 300000000 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "/tmp")
for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc9", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k] += 1
    names = sorted({c for v in agg.values() for c in v})
    print("kernel," + ",".join(names))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
        print(k + "," + ",".join("%.3g" % v.get(c, 0) for c in names))
    os.remove(f)
PY
