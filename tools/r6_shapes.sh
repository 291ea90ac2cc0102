# round 6: block shapes of swz_mdblock.hip (threads per block x cells per block) on the sparse levels of the 1 B run
cd $GRAFT_REPO_ROOT
for t in 64 256; do for b in 6 7 8 9; do
  echo "== threads $t bits $b: $(SWZ_SP_BLOCK_THREADS=$t SWZ_SP_BLOCK_BITS=$b SWZ_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --md-mode exact --also "" 2>&1 >/dev/null | grep "block path" | grep -o "level [0-9].*blocks of [0-9 x]* cells\|[0-9.]* ms$" | tr '\n' ' ')"
done; done
