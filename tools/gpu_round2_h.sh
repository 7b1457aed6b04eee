#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "level_tier or unbalanced or boundaries or config1 or determinism" 2>&1 | tail -15 ) > gpurun_out/h_tests.log 2>&1
tail -n 12 gpurun_out/h_tests.log
bash tools/ab.sh tools/libbvh_head.so bvh_amd/libbvh_mi355x.so 2
