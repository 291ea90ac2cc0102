# scratch script for one-off experiments through gpurun (`gpurun -- 'bash tools/exp.sh > gpurun_out/exp.log 2>&1'`);
# always bound what it runs: SWZ_MD_TIME_LIMIT (seconds per MIN_DISTANCE level) and `timeout` around every command
export SWZ_MD_TIME_LIMIT=20
SWZ_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep -E "MIN_DISTANCE level|ms_per_step" | cut -c1-230 | tail -30
