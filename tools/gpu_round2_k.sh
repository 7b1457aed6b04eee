#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_walk or config1 or config4" 2>&1 | tail -5 ) > gpurun_out/k_tests.log 2>&1
tail -n 3 gpurun_out/k_tests.log
bash tools/ab_multi.sh "$@"
