#!/bin/bash
# round 6, call K: closest hit of f64 batches through the guide walk — parity (incl. ties across items, grazing rays, out-of-range replay) and A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_k; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guide.py -x -q -k "triangle_stage or guide or fuzz or closest" 2>&1 | tail -8
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print(d['workload_name'], d['dtype'], d['value'], d['ms_per_step'], 'walk', p['traverse_kernel_ms'], 'trav', p['traverse_total_ms'], d['roofline']['kernel'], 'parity', d['parity']['equal'])"; }
for i in 1 2; do
echo "f64 closest, guide walk over items (default)"; run --dtype f64 --harness closest
echo "f64 closest, f64 walk over items (BVH_TUNE_14=0)"; BVH_TUNE_14=0 run --dtype f64 --harness closest
echo "f64 closest, f64 walk, one lane per ray (BVH_TUNE_14=0 BVH_TUNE_1=0)"; BVH_TUNE_14=0 BVH_TUNE_1=0 run --dtype f64 --harness closest
done 2>&1 | tee $O/closest_f64_guide_ab.log
