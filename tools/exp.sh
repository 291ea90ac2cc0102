SWZ_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --strategy FAST --points 200000000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/fast_sharded.out 2> gpurun_out/fast_sharded.err
echo rc=$?
tail -c 600 gpurun_out/fast_sharded.out; grep -v amdgpu.ids gpurun_out/fast_sharded.err | tail -15
