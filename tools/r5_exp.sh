# round-5 experiment runner: each line "tag|env|bench args" -> gpurun_out/r5/<tag>.json (+ .err)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
while IFS='|' read -r tag envs args; do
  [ -z "$tag" ] && continue
  env $envs timeout 600 python bench.py $args > gpurun_out/r5/$tag.json 2> gpurun_out/r5/$tag.err
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5/$tag.json | head -1) $(grep -o '"sample_min_distance": [0-9.]*' gpurun_out/r5/$tag.json | head -1)"
done
