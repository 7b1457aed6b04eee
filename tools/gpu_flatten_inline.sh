#!/bin/bash
# BVHGPU_TUNE_FLATTEN_INLINE (knob 21): the builder's wave tier writes the flatten's FLAT / WIDE parts of its subtrees.  Parity (the whole parity + fuzz files),
# then the headline and the 12 M-triangle entry with the knob on / off, alternating.   gpurun -- bash tools/gpu_flatten_inline.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_guide.py tests/test_gpu_host.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2 3; do
  for v in 1 0; do
    BVH_TUNE_21=$v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=json.load(open(d['detail'])) if 'phases_ms' not in d else d
print('inline=$v headline', d['value'], d['ms_per_step'], d['phases_ms'], (d.get('parity') or {}).get('equal'))"
  done
done
for v in 1 0 1 0; do
  BVH_TUNE_21=$v python bench.py --workload cubes12m --steps 10 --warmup 2 --settle-steps 2 --regions 3 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=json.load(open(d['detail'])) if 'phases_ms' not in d else d
print('inline=$v cubes12m', d['value'], d['ms_per_step'], d['phases_ms'], (d.get('parity') or {}).get('equal'))"
done
