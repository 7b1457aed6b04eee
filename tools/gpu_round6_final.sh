#!/bin/bash
# the validation call of round 6 with the round's final library: every GPU test, smoke, the default bench line as the driver runs it (compact line +
# detail file), the evidence profiles of every config, then a soak (fuzz over all queries with 400 seeds; build / flatten / CSR parity on 12 M triangles)
#   bash tools/gpu_round6_final.sh <tag>      → gpurun_out/final_*.log, gpurun_out/<tag>_bench_{default,detail}.json, gpurun_out/profiles_<tag>_*/
tag=${1:-r6_v2}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 2400 python -X faulthandler -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -30 ) > gpurun_out/final_tests.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1 )
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/${tag}_bench_detail.json > gpurun_out/${tag}_bench_default.json ) 2> gpurun_out/${tag}_bench_default.err
( timeout 2400 bash tools/gpu_round6_evidence.sh $tag > gpurun_out/final_profile.log 2>&1 )
( BVH_FUZZ_SEEDS=400 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k fuzz_all_queries 2>&1 | tail -4 ) > gpurun_out/final_fuzz_soak.log 2>&1
( timeout 900 python tools/big_scene_check.py 1000000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" ) > gpurun_out/final_big_scene.log 2>&1
tail -n 3 gpurun_out/final_tests.log; tail -n 1 gpurun_out/final_smoke.log; wc -c gpurun_out/${tag}_bench_default.json; head -c 300 gpurun_out/${tag}_bench_default.json; echo
tail -n 5 gpurun_out/${tag}_bench_default.err; tail -n 2 gpurun_out/final_fuzz_soak.log; cat gpurun_out/final_big_scene.log
