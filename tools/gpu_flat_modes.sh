#!/bin/bash
# A/B of the four ways the step can treat the FlatNode array (bench.py --flat-array): parity of every reader first, then 3 x 4 runs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r6_flat
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lazy_flatten" 2>&1 | grep -E "passed|failed"
for rep in 1 2 3; do for mode in lazy eager all beside; do
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity --no-excluded --flat-array $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', j['value'], j['ms_per_step'], j['regions_ms_per_step'], j['phases_ms'])"
done; done 2>&1 | tee gpurun_out/r6_flat/flat_modes_ab.log
for dt in f64; do for mode in lazy eager all; do
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity --no-excluded --dtype $dt --flat-array $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt $mode', j['value'], j['ms_per_step'], j['phases_ms'])"
done; done 2>&1 | tee -a gpurun_out/r6_flat/flat_modes_ab.log
