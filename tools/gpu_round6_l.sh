#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print(d['workload_name'], d['dtype'], d['value'], d['ms_per_step'], 'walk', p['traverse_kernel_ms'], 'trav', p['traverse_total_ms'], d['roofline']['kernel'], 'parity', d['parity']['equal'])"; }
for i in 1 2; do
echo "f64 index, guide"; run --dtype f64
echo "f64 closest, guide, noinline"; run --dtype f64 --harness closest
echo "f64 closest, guide, forceinline"; BVH_AMD_SO=$R/tools/libbvh_guide_inline.so run --dtype f64 --harness closest
echo "f32 closest"; run --harness closest
done
