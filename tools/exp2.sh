set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --sampler GRID_CENTER --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -2 | tee -a gpurun_out/exp2.jsonl
timeout 1200 python bench.py --steps 2 --warmup 1 2>&1 | tail -2 | tee -a gpurun_out/exp2.jsonl
rocm-smi --showmeminfo vram 2>&1 | tail -5
