cd $GRAFT_REPO_ROOT
echo default; timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE property 2>&1 | grep -E "Mpts" | cut -c1-100
echo block off; SWZ_MD_ROUNDS_BLOCK=0 timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE property 2>&1 | grep -E "Mpts" | cut -c1-100
echo blocks from 2 points per cell; SWZ_MD_ROUNDS_BLOCK_MIN_POP=2 timeout 600 python tools/clustered_probe.py 100000000 MIN_DISTANCE property 2>&1 | grep -E "Mpts" | cut -c1-100
bash tools/probe.sh tests tests/test_min_distance_property.py 2>&1 | tail -3
bash tools/probe.sh list < tools/lists/property_ab.txt 2>&1 | grep -v "^  property" | cut -c1-120
