#!/bin/bash
# round 6, call G: the flatten's second pass beside the walk (BVHGPU_TUNE_FLATTEN_LAZY = 2): parity, A/B, then the default bench
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_async_order.py tests/test_gpu_host.py -x -q -k "lazy or async or host or early" 2>&1 | tail -8
for rep in 1 2 3; do for mode in lazy eager beside; do
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity --no-excluded --flat-array $mode --detail-out $O/ab.json 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', j['value'], j['ms_per_step'], j['regions_ms_per_step'], j['phases_ms'])"
done; done 2>&1 | tee $O/flat_beside_ab.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.err; wc -c $O/bench_default.json
