#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/c_tests.log 2>&1
( timeout 600 python bench.py > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err )
( timeout 900 bash tools/profile_round.sh r2_v2 > gpurun_out/c_profile.log 2>&1 )
tail -n 5 gpurun_out/c_tests.log; head -c 600 gpurun_out/c_bench.json
