#!/bin/bash
# The two-launch level schedule's tile size (BVHGPU_TUNE_BUILD_LEVEL_TILE) and the block sums of many-tile items: builder parity tests, build time by
# tile with the node arrays compared (tools/level_tile_sweep.py), parity on 12 M triangles, the 12 M-triangle bench entry.
# gpurun -- bash tools/gpu_level_tile.sh  ->  gpurun_out/level_tile_sweep.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "level_tier or 1_2m or fuzz or lazy" 2>&1 | grep -E "passed|failed|error" | tail -2
python tools/level_tile_sweep.py 2>&1 | grep triangles | tee gpurun_out/level_tile_sweep.log
python tools/big_scene_check.py 1000000 2>&1 | grep -E "identical|levels" | tee -a gpurun_out/level_tile_sweep.log
ROUNDS=1 bash tools/ab_12m.sh - | tee -a gpurun_out/level_tile_sweep.log
