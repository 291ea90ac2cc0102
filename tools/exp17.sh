cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc17
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-include-regex "sp_neighbours|sp_round|md_gather" --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc17"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
with open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/exp17.txt", "w") as o:
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        o.write("%-50s fetch %.1f GB  hit %.3e miss %.3e  (n=%d)\n" % (k[:50], d.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0), cnt[(k, "FETCH_SIZE")]))
PY
rm -rf $OUT
