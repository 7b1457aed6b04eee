#!/bin/bash
# the last GPU call of round 5: every GPU test, smoke, the headline's profile round and the default bench line with the round's final library,
# then a soak (fuzz over all queries with 400 seeds; parity of build / flatten / CSR on 12 M triangles)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -30 ) > gpurun_out/final_tests.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1 )
( timeout 900 bash tools/gpu_round5_evidence.sh r5_v5 "c1" > gpurun_out/final_profile.log 2>&1 )
( timeout 900 python bench.py > gpurun_out/r5_v5_bench_default.json 2> gpurun_out/r5_v5_bench_default.err )
( BVH_FUZZ_SEEDS=400 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k fuzz_all_queries 2>&1 | tail -4 ) > gpurun_out/final_fuzz_soak.log 2>&1
( timeout 900 python tools/big_scene_check.py 1000000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" ) > gpurun_out/final_big_scene.log 2>&1
tail -n 3 gpurun_out/final_tests.log; tail -n 1 gpurun_out/final_smoke.log; head -c 200 gpurun_out/r5_v5_bench_default.json; echo; tail -n 2 gpurun_out/final_fuzz_soak.log; cat gpurun_out/final_big_scene.log
