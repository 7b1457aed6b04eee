#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15 ) > gpurun_out/j_tests.log 2>&1
tail -n 8 gpurun_out/j_tests.log
for i in 1 2; do for api in split fused; do
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --pipeline-streams 0 --no-extra --step-api $api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$api', d['value'], d['ms_per_step'], d['phases_ms'], d.get('parity',{}).get('equal'))"
done; done
