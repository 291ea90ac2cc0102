# round 6: the driver's default bench command with per-level debug output; prints the headline, the property leg and the also legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
SWZ_DEBUG=1 timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample ${CPU_SAMPLE:-0} > gpurun_out/r6/full.json 2> gpurun_out/r6/full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/full.json").read().strip().splitlines()[-1])
print("exact", d["ms_per_step"], d["value"], d["kernels_ms_per_step"], d["roofline"])
p = d.get("min_distance_property")
if p:
    print("property", p["ms_per_step"], p["kernels_ms_per_step"])
for k, v in (d.get("also") or {}).items():
    print(k, v.get("ms_per_step"), v.get("hbm_frac_end_to_end"), v.get("error"))
PY
grep -E "property|rounds|sweep" gpurun_out/r6/full.err | tail -${TAIL:-14} | cut -c1-260
