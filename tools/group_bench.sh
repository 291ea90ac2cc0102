# joint root sweep vs turns, 8 / 4 / 2 shards on the one device of the box
cd $GRAFT_REPO_ROOT
g++ -std=c++17 -O2 tools/group_bench.cpp -o /tmp/group_bench -Lschwarzwald_amd/lib -lswz_gpu -Wl,-rpath,$PWD/schwarzwald_amd/lib -Wl,-rpath,/opt/rocm/lib || exit 1
for sh in 8 2; do
  per=$((200000000 / sh))
  echo "== $sh shards x $per points, root swept by all shards at once"
  timeout 600 /tmp/group_bench $sh $per 2
  echo "== $sh shards x $per points, root in turns"
  SWZ_GROUP_JOINT_ROOT=0 timeout 600 /tmp/group_bench $sh $per 2
done
