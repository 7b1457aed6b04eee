#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not wide and not variants" 2>&1 | tail -15 ) > gpurun_out/e_tests.log 2>&1
( timeout 300 python bench.py --no-extra --no-cpu-baseline --no-parity --pipeline-streams 0 > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err )
( timeout 300 python bench.py --no-extra --no-cpu-baseline --no-parity --pipeline-streams 0 --dtype f64 > gpurun_out/e_bench64.json 2> gpurun_out/e_bench64.err )
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/e_trace -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; db=$(ls gpurun_out/e_trace/*.db gpurun_out/e_trace/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py $db gpurun_out/e_kernel_stats.md e > /dev/null; rm -rf gpurun_out/e_trace
tail -n 3 gpurun_out/e_tests.log; head -c 700 gpurun_out/e_bench.json
