# Copies the summaries of the last tools/profile_round.sh run (gpurun_out/profile/) into profiles/$1 (default r05) and derives
# traffic.json.  Nothing already in profiles/$1 is deleted: the new set is assembled in a temporary directory first and only
# then copied over what is there (files other tools left -- las_decode_*.txt, bin_backend_*.txt, extra probes -- stay).
set -euo pipefail
R=${1:-r06}
P=gpurun_out/profile
D=profiles/$R
[ -d "$P" ] || { echo "install_profiles: $P does not exist (run tools/profile_round.sh through gpurun first)" >&2; exit 1; }
[ -s "$P/source_sha16.txt" ] || { echo "install_profiles: $P/source_sha16.txt is missing -- the profile run did not finish" >&2; exit 1; }
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
shopt -s nullglob
for f in "$P"/bench_*.json "$P"/clustered_*.txt "$P"/fullsize_pytest.txt "$P"/fullsize_verification_1B.log \
         "$P"/group_joint_root_vs_turns.txt "$P"/pmc_*_by_kernel*.csv "$P"/source_sha16.txt "$P"/*.txt; do
  [ -f "$f" ] && cp "$f" "$T"/
done
if [ -f "$P/stats_run.json" ]; then cp "$P/stats_run.json" "$T/bench_under_rocprofv3.json"; fi
for pair in stats:bench_default stats_gc:GRID_CENTER stats_prop:property_mode stats_mb:1B_100batches_RANDOM_GRID \
            stats_mbmd:100batches_MIN_DISTANCE_FAST; do
  src=${pair%%:*}; name=${pair##*:}
  [ -f "$P/$src/bench_kernel_stats.csv" ] && cp "$P/$src/bench_kernel_stats.csv" "$T/rocprofv3_kernel_stats_$name.csv"
done
[ -n "$(ls -A "$T")" ] || { echo "install_profiles: nothing to install" >&2; exit 1; }
mkdir -p "$D"
# Files of another run must not end up under the new stamp: when $D was measured on other kernel sources its files are moved
# aside (profiles/$R/superseded_<their stamp>/) unless --keep-old is given.
if [ -s "$D/source_sha16.txt" ] && ! cmp -s "$D/source_sha16.txt" "$T/source_sha16.txt" && [ "${2:-}" != "--keep-old" ]; then
  old=$(tr -d '[:space:]' < "$D/source_sha16.txt")
  mkdir -p "$D/superseded_$old"
  find "$D" -maxdepth 1 -type f -exec mv {} "$D/superseded_$old/" \;
  echo "install_profiles: $D held a run on sources $old: moved to $D/superseded_$old/"
fi
cp "$T"/* "$D"/
if [ -f "$D/pmc_FETCH_SIZE_by_kernel.csv" ] && [ -f "$D/pmc_WRITE_SIZE_by_kernel.csv" ]; then python tools/make_traffic.py "$D"; else echo "(no PMC summaries yet: traffic.json not written)"; fi
echo "run on sources $(cat "$D/source_sha16.txt"), tree has $(python -c 'import bench; print(bench.library_source_sha16())')"
