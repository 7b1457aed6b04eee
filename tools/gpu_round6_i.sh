#!/bin/bash
# round 6, call I: closest hit over items in f64 ((ray, item) slots) — parity incl. ties across items, A/B against one lane per ray; configs[2] closest with items forced
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "triangle_stage or fuzz or closest" 2>&1 | tail -5
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print(d['workload_name'], d['dtype'], d['value'], d['ms_per_step'], 'walk', p['traverse_kernel_ms'], 'trav', p['traverse_total_ms'], d['roofline']['kernel'], 'parity', d['parity']['equal'])"; }
for i in 1 2; do
echo "f64 closest, items (default)"; run --dtype f64 --harness closest
echo "f64 closest, one lane per ray (BVH_TUNE_1=0)"; BVH_TUNE_1=0 run --dtype f64 --harness closest
echo "f32 closest, items (default)"; run --harness closest
echo "f32 closest, one lane per ray"; BVH_TUNE_1=0 run --harness closest
echo "configs[2] closest, default (whole rays)"; run --workload standin-primary --harness closest --steps 30
echo "configs[2] closest, items forced (BVH_TUNE_1=2)"; BVH_TUNE_1=2 run --workload standin-primary --harness closest --steps 30
done 2>&1 | tee $O/closest_items_ab.log
