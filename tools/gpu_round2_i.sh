#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -x -q 2>&1 | tail -15 ) > gpurun_out/i_tests.log 2>&1
tail -n 6 gpurun_out/i_tests.log
bash tools/ab.sh tools/libbvh_head.so bvh_amd/libbvh_mi355x.so 2
