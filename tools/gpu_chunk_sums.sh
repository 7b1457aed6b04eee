#!/bin/bash
# round 6: block sums for many-tile items in the two-launch level schedule (k_bin / k_split, scenes above 250 k shapes) — parity with the block size forced
# down to 4 tiles on the builder tests, parity at 1.2 M and 12 M shapes, build time before / after
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r6_chunk
[ -f $R/tools/libbvh_chunk4.so ] || python bvh_amd/build_ext.py --variant $R/tools/libbvh_chunk4.so BVH_CHUNK_TILES=4 > /dev/null 2>&1
echo "== builder parity tests, block = 4 tiles (every item above 2048 shapes takes the new path)"
BVH_AMD_SO=$R/tools/libbvh_chunk4.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "level_tier or large_scene or unbalanced or mid_tier or degenerate or parity_sizes or fuzz" 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== the same, in-tree library"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "level_tier or large_scene or unbalanced" 2>&1 | grep -E "passed|failed|Error" | tail -3
for n in 100000 1000000; do
echo "== big_scene_check $n cubes"; timeout 900 python tools/big_scene_check.py $n 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
done
echo "== bench cubes12m"
python bench.py --workload cubes12m --steps 20 --warmup 3 --settle-steps 5 --no-cpu-baseline --pipeline-streams 0 --no-extra --parity-max-rays 1000000 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['phases_ms'], j['parity'])"
