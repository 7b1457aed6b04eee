#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide or variants or config1 or config4 or config0 or full_size or c_abi" 2>&1 | tail -15 ) > gpurun_out/d_tests.log 2>&1
python tools/wide_prof.py 1000000 2 > gpurun_out/d_wp2.log 2>&1
python tools/wide_prof.py 1000000 1 > gpurun_out/d_wp1.log 2>&1
python tools/wide_prof.py 1000000 0 > gpurun_out/d_wp0.log 2>&1
( timeout 300 python tools/wide_sweep.py f32 1000000 > gpurun_out/d_sweep_f32.log 2>&1 )
( timeout 300 python bench.py --no-extra --no-cpu-baseline > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err )
tail -n 3 gpurun_out/d_tests.log; head -c 900 gpurun_out/d_bench.json
