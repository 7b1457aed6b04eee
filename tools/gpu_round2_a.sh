#!/bin/bash
# first GPU call of round 2: tests (new ones first), bench, wide sweep, profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/a_tests_new.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scene.py -x -q -m gpu 2>&1 | tail -40 ) > gpurun_out/a_tests_parity.log 2>&1
( timeout 600 python bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err )
( timeout 300 python tools/wide_sweep.py f32 1000000 > gpurun_out/a_sweep_f32.log 2>&1 )
( timeout 200 python tools/wide_sweep.py f64 1000000 > gpurun_out/a_sweep_f64.log 2>&1 )
( timeout 200 python tools/wide_sweep.py f32 8000000 > gpurun_out/a_sweep_f32_8m.log 2>&1 )
( timeout 900 bash tools/profile_round.sh r2_v1 > gpurun_out/a_profile.log 2>&1 )
tail -5 gpurun_out/a_tests_new.log gpurun_out/a_tests_parity.log; head -c 1500 gpurun_out/a_bench.json; tail -3 gpurun_out/a_bench.err
