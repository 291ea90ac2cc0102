# One entry point for the measurements made while tuning (run through gpurun: `gpurun --timeout 1800 -- 'bash tools/probe.sh <what> ...'`).
# Everything lands in gpurun_out/probe/.  Boxes of the pool differ by +-4 %: compare variants inside ONE call.
#
#   tests [pytest args]      GPU tests, default the whole -m gpu suite
#   list < file              lines "tag|ENV=.. ENV=..|bench.py args": one bench.py run each -> <tag>.json (+ .err), ms_per_step printed
#                            (a tag that starts with sleep<seconds>_ waits first: the driver wipes the memory a process frees)
#   levels [bench args]      bench.py with SWZ_DEBUG=1: the library's per-level lines (block path, sweeps, rounds, the incremental subset)
#   variants [bench args]    the same with the default library and every schwarzwald_amd/lib/libswz_v*.so (tools/build_variant.sh;
#                            -DSWZ_SB_STATS builds print the block kernel's phase times); VAR_ENV="SWZ_SP_BLOCK_DBG=8" passes switches on
#   multibatch [orders]      1 B points in 100 batches, MIN_DISTANCE FAST exact (the reference's default operating point), per order
#                            (default "tiles uniform"); MB_ENVS="default SWZ_SP_INCREMENTAL=0" runs every order under each environment
#   stats [bench args]       rocprofv3 --kernel-trace --stats of one bench.py command: the top of the kernel summary
#   configs                  BASELINE configs 4 / 5 as dry runs on ONE GPU (8 ranks / 8 shards)
cd $GRAFT_REPO_ROOT
O=gpurun_out/probe
mkdir -p $O
what=${1:-tests}
[ $# -gt 0 ] && shift
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1], "no bench line:", e)
    sys.exit(0)
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "first_data_set_ms", d.get("first_data_set_ms"), "V", d.get("visit_factor"),
      {k: round(v, 1) for k, v in (d.get("kernels_ms_per_step") or {}).items()})
p = d.get("min_distance_property")
if p:
    print("  property", p["ms_per_step"], {k: round(v, 1) for k, v in p["kernels_ms_per_step"].items()})
for k, v in (d.get("also") or {}).items():
    print("  also", k, v.get("ms_per_step"), v.get("error"))
PY
}
case $what in
tests)
  if [ $# -eq 0 ]; then set -- tests; fi
  timeout 2300 python -m pytest "$@" -x -q -m gpu --durations=5 2>&1 | tail -25 ;;
list)
  while IFS='|' read -r tag envs args; do
    [ -z "$tag" ] && continue
    case $tag in sleep*) t=${tag#sleep}; sleep ${t%%_*} ;; esac   # (a tag "sleep20_..." waits that long first)
    env $envs timeout 900 python bench.py $args > $O/$tag.json 2> $O/$tag.err
    line $O/$tag.json
  done ;;
levels)
  SWZ_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --also "" "$@" > $O/levels.json 2> $O/levels.err
  grep -E "block path|sparse path|sweep|rounds|can change|thread 0" $O/levels.err | tail -${TAIL:-24} | cut -c1-420
  line $O/levels.json ;;
variants)
  for lib in schwarzwald_amd/lib/libswz_gpu.so schwarzwald_amd/lib/libswz_v*.so; do
    [ -f "$lib" ] || continue
    echo "== $lib $VAR_ENV"
    env $VAR_ENV SWZ_GPU_LIBRARY=$PWD/$lib SWZ_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --also "" "$@" 2>&1 >$O/variant.json \
      | grep -E "block path|thread 0|property level" | tail -${TAIL:-4} | cut -c1-60,150-500
    line $O/variant.json
  done ;;
multibatch)
  for order in ${@:-tiles uniform}; do
    IFS=';' read -ra envs <<< "${MB_ENVS:-A=1}"
    for e in "${envs[@]}"; do
      f=$O/mb_${order}_$(echo "$e" | tr -c 'A-Za-z0-9\n' '_').json
      env $e timeout 900 python bench.py --batches 100 --strategy ${MB_STRATEGY:-FAST} --batch-order $order --md-mode ${MB_MODE:-exact} --steps 2 --warmup 1 --cpu-sample 0 > $f 2> ${f%.json}.err
      echo "-- $order $e"
      line $f
    done
  done ;;
stats)
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/$O/stats && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o p -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --also "" "$@" > $GRAFT_REPO_ROOT/$O/stats.json 2> $GRAFT_REPO_ROOT/$O/stats.err )
  f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/stats_kernel_stats.csv && head -${TOP:-30} "$f" | awk -F'","' '{print substr($1,2,90), $2, $3, $4}'
  rm -rf $O/stats
  line $O/stats.json ;;
configs)
  run() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; line $O/$tag.json; tail -2 $O/$tag.err | cut -c1-300; }
  run config4_dryrun_8ranks_1gpu_200M --config 4 --gpus 8 --one-device --total-points 200000000 --steps 2 --warmup 1
  run config5_dryrun_8ranks_1gpu_200M --config 5 --gpus 8 --one-device --total-points 200000000 --steps 1 --warmup 1
  run config4_dryrun_group_8shards_1gpu_200M --config 4 --gpus 8 --driver group --group-devices 1 --total-points 200000000 --steps 2 --warmup 1
  run config5_dryrun_group_8shards_1gpu_200M --config 5 --gpus 8 --driver group --group-devices 1 --total-points 200000000 --steps 2 --warmup 1 ;;
*)
  sed -n 1,16p tools/probe.sh ;;
esac
