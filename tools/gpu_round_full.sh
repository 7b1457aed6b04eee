#!/bin/bash
# full validation of a round on the GPU box: every GPU test, smoke, the default bench line, the per-config evidence profiles
#   bash tools/gpu_round_full.sh <tag>      → gpurun_out/full_*.{log,json}, gpurun_out/profiles_<tag>_{c1,c1closest,c1tris,c2,c2closest,c3,c4,c4f64,c12m}/
tag=${1:-r5_v3}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -45 ) > gpurun_out/full_tests.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.log 2>&1 )
( timeout 900 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err )
( timeout 1800 bash tools/gpu_round5_evidence.sh $tag > gpurun_out/full_profile.log 2>&1 )
tail -n 4 gpurun_out/full_tests.log; tail -n 2 gpurun_out/full_smoke.log; head -c 400 gpurun_out/full_bench.json
